#!/bin/bash
# Round profile: rocprofv3 kernel trace + separate PMC passes of the bench command (DBs stay in /tmp on the box,
# only the summaries land in gpurun_out/).   usage: bash scripts/profile_round.sh r01d
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import torch" >/dev/null 2>&1
cd $R
TAG=${1:-r01d}
# 1) headline workload only, single-stream schedule: the bench line's roofline (instrumented passes on one stream) must agree
#    with this table.  The default schedule (trunk on its own stream: kernels of the two streams share the chip, so a
#    kernel's duration is no longer its own) goes into ${TAG}_kernel_stats_two_streams.txt.
PNMN_NMN_STREAM=0 timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o $TAG -- python bench.py --no-cpu-baseline --no-extras > gpurun_out/${TAG}_prof_bench.log 2>&1
python profiles/summarize.py $(find /tmp/prof_$TAG -name '*_results.db' | head -1) > gpurun_out/${TAG}_kernel_stats.txt 2>&1
grep '^{' gpurun_out/${TAG}_prof_bench.log > gpurun_out/${TAG}_prof_bench.json
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof2_$TAG -o $TAG -- python bench.py --no-cpu-baseline --no-extras --no-roofline > gpurun_out/${TAG}_prof2_bench.log 2>&1
python profiles/summarize.py $(find /tmp/prof2_$TAG -name '*_results.db' | head -1) > gpurun_out/${TAG}_kernel_stats_two_streams.txt 2>&1
# 2) HBM traffic counters, one counter per pass (MI355X_MICROARCH.md: no mixing with other trace domains)
for C in FETCH_SIZE WRITE_SIZE; do
  # (with the roofline passes ON: the run's own bench line counts the conv launches of its instrumented passes -- the last
  #  conv launches of the process -- and their algorithmic bytes; summarize.py sums the counters of exactly those)
  PNMN_NMN_STREAM=0 timeout 900 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc_$C -o p -- python bench.py --steps 2 --warmup 1 --roofline-passes 6 --no-cpu-baseline --no-extras > gpurun_out/${TAG}_pmc_$C.log 2>&1
  python profiles/summarize.py --pmc $(find /tmp/pmc_$C -name '*_results.db' | head -1) gpurun_out/${TAG}_pmc_$C.log > gpurun_out/${TAG}_pmc_$C.txt 2>&1
done
# 3) MFMA utilisation (SQ + GRBM counters share a pass)
PNMN_NMN_STREAM=0 timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmc_mfma -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-roofline > gpurun_out/${TAG}_pmc_MFMA.log 2>&1
python profiles/summarize.py --mfma $(find /tmp/pmc_mfma -name '*_results.db' | head -1) > gpurun_out/${TAG}_pmc_MFMA.txt 2>&1
cut -c1-400 gpurun_out/${TAG}_prof_bench.json
# 4) the timed steps alone, single-stream schedule: per-iteration table of the last 20 iterations (= the timed region when
#    nothing runs behind it), whose conv launch durations are what bench.py's instrumented passes must agree with -- the
#    whole-process table of 1) also averages the settle / warm-up steps, whose generator samples fewer valid programs
PNMN_NMN_STREAM=0 timeout 900 rocprofv3 --kernel-trace -d /tmp/prof3_$TAG -o $TAG -- python bench.py --no-cpu-baseline --no-extras --no-roofline > gpurun_out/${TAG}_prof3_bench.log 2>&1
python profiles/summarize.py --steady 20 $(find /tmp/prof3_$TAG -name '*_results.db' | head -1) > gpurun_out/${TAG}_steady_single_stream.txt 2>&1
# 5) the recurrent kernels alone at the headline shapes (bench.py recurrent_kernel_report): HBM counters of exactly those launches
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc_rec_$C -o p -- python scripts/r06_recurrent_pmc.py > gpurun_out/${TAG}_pmc_recurrent_$C.log 2>&1
  python profiles/summarize.py --pmc $(find /tmp/pmc_rec_$C -name '*_results.db' | head -1) > gpurun_out/${TAG}_pmc_recurrent_$C.txt 2>&1
done
