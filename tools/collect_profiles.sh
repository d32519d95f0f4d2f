#!/bin/bash
# Collects the round's tracked profiles on a GPU box:  bash tools/collect_profiles.sh r02
# Every rocprofv3 invocation has the program itself after `--` and never mixes --pmc with a trace domain.  Raw output goes to
# gpurun_out/<tag>_*/ (scratch), the reductions to gpurun_out/<tag>_*.txt; copy those into profiles/ afterwards.
set -e -o pipefail
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp

echo "[collect] bench lines"
for cfg in teacher student probe; do
  python3 $R/bench.py --config $cfg > $O/${TAG}_bench_$cfg.json 2> $O/${TAG}_bench_$cfg.err
  tail -c 400 $O/${TAG}_bench_$cfg.json; echo
done
python3 $R/bench.py --stress --no-cpu-baseline > $O/${TAG}_bench_stress.json 2> $O/${TAG}_bench_stress.err      # configs[4] shapes at its per-GPU batch (B 32)
tail -c 400 $O/${TAG}_bench_stress.json; echo

for cfg in teacher student; do
  echo "[collect] kernel trace $cfg"
  CMD="python3 $R/bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --no-hbm-table"
  rm -rf $O/${TAG}_kt_$cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_kt_$cfg -- $CMD > $O/${TAG}_kt_$cfg.log 2>&1
  {
    echo "# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --no-hbm-table   ($TAG, final state)"
    echo "# NB: under the profiler the captured graph's branches are serialised (one queue), so this file gives kernel DURATIONS with the"
    echo "#     chip to themselves at profiler clocks, not the overlap; bench.py's in-step clocks (roofline.achieved) are read inside the kernels."
    echo "# bench line of this run: $(grep '^{' $O/${TAG}_kt_$cfg.log | tail -1 | cut -c1-400)"
    python3 $R/tools/prof_summary.py $O/${TAG}_kt_$cfg 70
    if [ $cfg = teacher ]; then python3 $R/tools/prof_gemm_agreement.py $O/${TAG}_kt_$cfg 240; fi
  } > $O/${TAG}_kerneltrace_bench_$cfg.txt
  if [ $cfg = teacher ]; then      # the trace's GEMM mean and the SAME run's in-kernel launch clocks side by side (bench.py reads it back)
    python3 $R/tools/make_profile_json.py gemm $O/${TAG}_kt_$cfg $O/${TAG}_kt_$cfg.log $O/${TAG}_rocprof_gemm.json $TAG
    python3 $R/tools/prof_by_grid.py $O/${TAG}_kt_$cfg gemm_bf16_nt_v > $O/${TAG}_gemm_by_grid.txt 2>/dev/null || true
  fi
  head -12 $O/${TAG}_kerneltrace_bench_$cfg.txt
  rm -rf $O/${TAG}_kt_$cfg
done

echo "[collect] PMC FETCH_SIZE / WRITE_SIZE (separate passes, eager step)"
PCMD="python3 $R/bench.py --steps 3 --warmup 1 --eager --no-cpu-baseline --no-hbm-table"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/${TAG}_pmc_$c
  rocprofv3 --pmc $c --output-format csv -d $O/${TAG}_pmc_$c -- $PCMD > $O/${TAG}_pmc_$c.log 2>&1
done
{
  echo "# rocprofv3 --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2), no trace domains: python3 bench.py --steps 3 --warmup 1 --eager --no-cpu-baseline --no-hbm-table   ($TAG, final state)"
  echo "# columns: dispatches, total KiB, mean KiB per dispatch, kernel   (FETCH_SIZE is doubled in traffic.json as the guide prescribes for gfx950)"
  echo "## FETCH_SIZE"; python3 $R/tools/pmc_summary.py $O/${TAG}_pmc_FETCH_SIZE FETCH_SIZE 24
  echo "## WRITE_SIZE"; python3 $R/tools/pmc_summary.py $O/${TAG}_pmc_WRITE_SIZE WRITE_SIZE 24
} > $O/${TAG}_pmc_fetch_write_bench_teacher.txt
head -8 $O/${TAG}_pmc_fetch_write_bench_teacher.txt
cp $R/profiles/traffic.json $O/${TAG}_traffic.json 2>/dev/null || true
python3 $R/tools/make_profile_json.py traffic $O/${TAG}_pmc_FETCH_SIZE $O/${TAG}_pmc_WRITE_SIZE $O/${TAG}_traffic.json $TAG
rm -rf $O/${TAG}_pmc_FETCH_SIZE $O/${TAG}_pmc_WRITE_SIZE
echo "[collect] PMC SQ counters: DuETT embedding stage, CXR attention"
SQE="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_LDS"
rm -rf $O/${TAG}_pmc_embed
rocprofv3 --pmc $SQE --output-format csv -d $O/${TAG}_pmc_embed -- python3 $R/tools/prof_embed.py 3 > $O/${TAG}_pmc_embed.log 2>&1
{
  echo "# rocprofv3 --pmc $SQE -- python3 tools/prof_embed.py 3   ($TAG; medp_duett_embed_fwd at B 64, T 96, V 48; mean per dispatch)"
  python3 $R/tools/pmc_all.py $O/${TAG}_pmc_embed embed
} > $O/${TAG}_pmc_embed.txt
rm -rf $O/${TAG}_pmc_embed
SQA="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
rm -rf $O/${TAG}_pmc_attn
rocprofv3 --pmc $SQA --output-format csv -d $O/${TAG}_pmc_attn -- python3 $R/tools/bench_attn.py > $O/${TAG}_pmc_attn.log 2>&1
{
  echo "# rocprofv3 --pmc $SQA -- python3 tools/bench_attn.py   ($TAG; attn_fwd_dh64_kernel at B 64, S 257, H 12; mean per dispatch)"
  python3 $R/tools/pmc_all.py $O/${TAG}_pmc_attn attn_fwd
} > $O/${TAG}_pmc_attention.txt
rm -rf $O/${TAG}_pmc_attn
cat $O/${TAG}_pmc_embed.txt $O/${TAG}_pmc_attention.txt
echo "[collect] branch timing (tools/time_branches.py)"
{
  echo "# python3 tools/time_branches.py [student]   ($TAG; HBM-resident batch; whole captured step vs each branch replayed alone; in-kernel launch clocks of the 48 block GEMMs)"
  python3 $R/tools/time_branches.py 2>/dev/null | grep -E "GEMMs|whole step"
  python3 $R/tools/time_branches.py student 2>/dev/null | grep -E "GEMMs|whole step"
} > $O/${TAG}_branch_times.txt
cat $O/${TAG}_branch_times.txt
{
  echo "# MEDP_ATTN_FEWQ=0 / default: python3 tools/time_attn_small.py   ($TAG; perceiver attention cores, isolated launches incl. ~10 us of Python per call)"
  MEDP_ATTN_FEWQ=0 python3 $R/tools/time_attn_small.py 2>/dev/null | grep FEWQ
  python3 $R/tools/time_attn_small.py 2>/dev/null | grep FEWQ
} > $O/${TAG}_attn_few_query.txt
cat $O/${TAG}_attn_few_query.txt
echo "[collect] done"
