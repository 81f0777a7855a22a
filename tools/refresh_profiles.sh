#!/bin/bash
# Runs on the GPU box (through gpurun): the bench line, the rocprofv3 kernel statistics of the same command, the two
# PMC passes for HBM traffic and two SQ passes for the chain kernels' issue / LDS figures; everything lands under
# gpurun_out/refresh/.  Afterwards, in the container: python3 tools/make_profiles.py r04  (summaries -> profiles/).
set -e
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/refresh   # (delete the local copy first: gpurun merges new files into it)
PART=${1:-all}            # a: the bench line and the profiler passes; b: sweeps, shapes, rANS 4x8, host rates (two gpurun calls)
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
if [ "$PART" != b ]; then
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python3 $R/bench.py --no-cpu --no-host --no-configs4 --no-hetero > $O/bench_prof.json 2> $O/bench_prof.err
P="--steps 1 --warmup 1 --no-cpu --no-host --no-configs4 --no-hetero"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch --output-format csv -- python3 $R/bench.py $P > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write --output-format csv -- python3 $R/bench.py $P > $O/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY \
    -d $O/sq1 --output-format csv -- python3 $R/bench.py $P > $O/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
    -d $O/sq2 --output-format csv -- python3 $R/bench.py $P > $O/sq2.log 2>&1 || echo "sq2 pass failed (counter set not available)"
# keep the merge small: only the CSVs the summaries need
find $O -name "*agent_info.csv" -delete
tail -n 1 $O/bench.json
fi
if [ "$PART" = a ]; then exit 0; fi
# round 4: batches whose blocks differ in length, alphabet and order, with the scheduling switched on step by step
HETERO_GIB=16 python3 $R/tools/batch_sweep.py --hetero $O/hetero.jsonl > $O/hetero.log 2>&1
HETERO_GIB=4 python3 $R/tools/batch_sweep.py --hetero $O/hetero_4g.jsonl > $O/hetero_4g.log 2>&1
REPS=3 R4X16_SCHED_TRACE=1 python3 $R/tools/hetero_trace.py 16 > $O/hetero_sched_trace.txt 2>&1
# round 3: the batch-size sweep (what a caller gets below one round of resident streams) and the other shapes
python3 $R/tools/batch_sweep.py $O/batch_sweep.jsonl > $O/batch_sweep.log 2>&1
bash $R/tools/shapes.sh > $O/shapes.jsonl 2> $O/shapes.err
bash $R/tools/sq_probe.sh > $O/sq_small.txt 2>&1; cp $R/gpurun_out/sqp/summary.txt $O/sq_small_summary.txt
tail -n 3 $O/batch_sweep.jsonl
# round 4: rANS 4x8 on the 4x16 loops - one full round of ITS resident streams (30 per CU), the 4x16 round, two rounds
for n in 7680 11520 15360; do python3 $R/tools/rate_4x8.py $n 2>/dev/null | grep rANS; done > $O/rate_4x8.txt
rocprofv3 --kernel-trace --stats -d $O/x8stats --output-format csv -- python3 $R/tools/rate_4x8.py 11520 > $O/x8stats.log 2>&1
find $O -name "*agent_info.csv" -delete
cat $O/rate_4x8.txt
# host-buffer batches (PCIe both ways): the probe at three sizes, the reference's own -t loop through the CLI
for n in 3072 4096 8192; do python3 $R/tools/host_probe.py $n 2>/dev/null | tail -n 1; done > $O/host_probe.txt
bash $R/tools/host_batch_rate.sh 3.9 1 > $O/host_batch_rate.txt 2>&1
cat $O/host_probe.txt; tail -n 4 $O/host_batch_rate.txt
