#!/bin/bash
# Collects the round's profile evidence on the GPU box (run through gpurun):
#   1) rocprofv3 --kernel-trace --stats of bench.py       -> gpurun_out/prof/trace
#   2) separate --pmc passes (FETCH_SIZE / WRITE_SIZE)     -> gpurun_out/prof/fetch, gpurun_out/prof/write
# rocprofv3 is given the python program directly (no env/bash wrappers after `--`).
# Usage: profile_round.sh [subdir [extra bench.py args ...]]   e.g.  profile_round.sh prof_bf16 --mfma bf16
R=${GRAFT_REPO_ROOT:-/root/repo}
SUB=${1:-prof}
[ $# -gt 0 ] && shift
O=$R/gpurun_out/$SUB
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-isolated-pass --no-secondary --warm-seconds 0 --sustained-steps 0 "$@" > $O/bench_under_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-isolated-pass --no-secondary --warm-seconds 0 --sustained-steps 0 "$@" > $O/bench_under_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-isolated-pass --no-secondary --warm-seconds 0 --sustained-steps 0 "$@" > $O/bench_under_write.log 2>&1
# 3) the shader clock each kernel HELD: GRBM_GUI_ACTIVE / 8 XCDs / the dispatch's duration (MI355X_MICROARCH.md, 'DVFS give-back'), its own pass
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/clock -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-isolated-pass --no-secondary --warm-seconds 0 --sustained-steps 0 "$@" > $O/bench_under_clock.log 2>&1
# 4) (SAR_VALU_PASS=1: the pad250 leg) vector-ALU instruction counts of the radar kernels, their own pass (tools/summarize_valu.py)
if [ -n "$SAR_VALU_PASS" ]; then
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/valu -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-isolated-pass --no-secondary --warm-seconds 0 --sustained-steps 0 "$@" > $O/bench_under_valu.log 2>&1
fi
cd $R && python3 bench.py --steps 10 --warmup 3 --no-secondary "$@" > $O/bench_plain.log 2>&1
tail -1 $O/bench_plain.log | cut -c1-400
ls $O/*/*/ | head -20
