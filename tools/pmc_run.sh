#!/bin/bash
# rocprofv3 counter passes for the given workloads (default c2 c3 c5), whole crowd on one GPU, summarised into
# gpurun_out/<tag>_pmc_summary.csv (copy to profiles/ to have it judged).  Counters only: no trace domains beside --pmc.
#   bash tools/pmc_run.sh r02 c2 c5
tag=${1:-r04}; shift
wl=${@:-c2 c3 c5}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
# the counters are those of the pedestrian-pair kernel ON ITS OWN (what bench.py times alone with HIP events): the geometry
# workgroups stay in a launch of their own for these passes
export SFM_PAIR_GEO=0
specs=""
for w in $wl; do
  t=40; [ $w = c5 ] && t=12
  mkdir -p $out/$w; export PMC_META=$out/$w/meta.json
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES SQ_INSTS_SALU GRBM_GUI_ACTIVE \
      --output-format csv -d $out/$w/sq -o p -- python3 $root/tools/pmc_ticks.py $w $t > $out/${w}_sq.log 2>&1 || exit 1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/$w/fetch -o p -- python3 $root/tools/pmc_ticks.py $w $t > $out/${w}_fetch.log 2>&1 || exit 1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/$w/write -o p -- python3 $root/tools/pmc_ticks.py $w $t > $out/${w}_write.log 2>&1 || exit 1
  # the fp32 instruction mix (round 4: packed instructions make one instruction of two -- SQ_INSTS_VALU alone no longer says how much
  # arithmetic was issued); a pass of its own, and not fatal if this box does not know the counters
  rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $out/$w/mix -o p -- python3 $root/tools/pmc_ticks.py $w $t > $out/${w}_mix.log 2>&1 || echo "  (no instruction-mix pass for $w)"
  specs="$specs $w=$out/$w"
  echo "pmc passes done: $w"
done
python3 $root/tools/pmc_summarize.py $root/gpurun_out/${tag}_pmc_summary.csv $specs
