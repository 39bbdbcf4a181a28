#!/bin/bash
# VALU instructions and busy cycles of one replayed rank's pair launch with and without the border / obstacle workgroups in it
#   bash tools/r04_shard_geo_pmc.sh [rank]
r=${1:-1}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/shard_geo_pmc
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for f in ${FORCE_SETS:-all ped}; do
  export PROBE_FORCES=$f
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $out/$f -o p -- python3 $root/tools/shard_trace.py $r plain > $out/$f.log 2>&1 || exit 1
  python3 - $out/$f/p_counter_collection.csv $f <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    if 'pair' in k or 'epilogue' in k or 'geometry' in k:
        print(sys.argv[2], k, {c: round(sum(x[len(x)//2:])/max(1,len(x[len(x)//2:]))) for c,x in v.items()})
PY
done
