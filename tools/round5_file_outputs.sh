#!/bin/bash
# Round 5: what real files allow (GPU box, repository root).  (a) the box's file systems under N writers, buffered and O_DIRECT after fallocate;
# (b) --output-sequences-separate with ONE GPU context and a pool of writers over rows the sink may keep (v2m_splice_rows_held), per pool size.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/file_outputs; mkdir -p $OUT
python3 -c 'import __graft_entry__ as g; g.build()' > /dev/null
{
for d in /dev/shm /tmp; do
	mkdir -p $d/v2m_probe
	python3 tools/file_write_probe.py $d/v2m_probe 16
	python3 tools/file_write_probe.py $d/v2m_probe 16 --direct
	rmdir $d/v2m_probe
done
} > $OUT/file_write_probe.txt 2>&1
{
for d in /dev/shm /tmp; do
	for w in 1 2 4 8 12; do
		echo "== V2M_WRITER_THREADS=$w, one context, $d"
		V2M_WRITER_THREADS=$w python3 tools/e2e_separate_files.py $d 2>&1 | grep -v amdgpu.ids
	done
	echo "== V2M_WRITER_THREADS=8 V2M_HELD_SLOTS=8, one context, $d"
	V2M_WRITER_THREADS=8 V2M_HELD_SLOTS=8 python3 tools/e2e_separate_files.py $d 2>&1 | grep -v amdgpu.ids
	echo "== V2M_WRITER_THREADS=8, --device=0,0, $d"
	V2M_WRITER_THREADS=8 python3 tools/e2e_separate_files.py $d 0,0 2>&1 | grep -v amdgpu.ids
done
} > $OUT/separate_files.txt 2>&1
cat $OUT/file_write_probe.txt $OUT/separate_files.txt
