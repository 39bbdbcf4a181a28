#!/bin/bash
# VALU instructions per launch of the list-mode pair kernel at c3 / c5 for the libraries named (whole crowd, SFM_PAIR_GEO=0 as in pmc_run.sh)
#   bash tools/r04_pair_valu.sh "" variants/libsfm_head.so
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/pair_valu
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export SFM_PAIR_GEO=0
n=0
for l in "$@"; do
  n=$((n+1))
  if [ -n "$l" ]; then export SFM_LIB_PATH=$root/$l; else unset SFM_LIB_PATH; fi
  for w in c3 c5; do
    t=40; [ $w = c5 ] && t=12
    timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $out/$n$w -o p -- python3 $root/tools/pmc_ticks.py $w $t > $out/$n$w.log 2>&1 || exit 1
    python3 - $out/$n$w/p_counter_collection.csv "${l:-product} $w" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    if 'pair_sym' in k:
        print(sys.argv[2], {c: round(sum(x[len(x)//2:])/max(1,len(x[len(x)//2:]))) for c,x in v.items()})
PY
  done
done
