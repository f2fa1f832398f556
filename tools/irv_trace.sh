#!/bin/bash
# kernel trace of three frames (synthetic, then real content): per-kernel durations of the refinement stage
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=${1:-gpurun_out/irv_trace}
mkdir -p $OUT
rocprofv3 --kernel-trace -d $OUT/s -o f --output-format csv -- python3 tools/frame_loop.py 3 > $OUT/s.log 2>&1
rocprofv3 --kernel-trace -d $OUT/r -o f --output-format csv -- python3 tools/real_frame_loop.py 3 > $OUT/r.log 2>&1
python3 - <<PY
import csv,glob
for tag in ("s","r"):
    f=glob.glob("$OUT/%s/**/f_kernel_trace.csv" % tag, recursive=True)[0]
    rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
    seq=[(r['Kernel_Name'].split('(')[0].replace('void ','').replace('stm::','')[:28], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3) for r in rows]
    idx=[i for i,s in enumerate(seq) if 'demux' in s[0]][2]
    print(tag, " ".join("%s=%.1f" % (n,d) for n,d in seq[idx:idx+26]))
PY
