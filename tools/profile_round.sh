#!/bin/bash
# Profiles of the default bench command for profiles/: kernel trace + stats, then
# one counter pass per run (HBM read bytes, HBM write bytes, wave/VALU activity).
# usage: tools/profile_round.sh r01     (run on the GPU box through gpurun)
set -u
tag=${1:-r01}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/profile_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- python3 $root/bench.py > $out/bench_under_rocprof.json 2> $out/bench_under_rocprof.err || exit 1
short="--steps 1 --warmup 1 --cpu-seconds 0 --no-stats-pass"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o c -- python3 $root/bench.py $short > /dev/null 2> $out/pmc_fetch.err || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o c -- python3 $root/bench.py $short > /dev/null 2> $out/pmc_write.err || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_sq -o c -- python3 $root/bench.py $short > /dev/null 2> $out/pmc_sq.err || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS --output-format csv -d $out/pmc_insts -o c -- python3 $root/bench.py $short > /dev/null 2> $out/pmc_insts.err || exit 1
python3 $root/tools/pmc_table.py $out/pmc_insts/c_counter_collection.csv > $out/insts_table.md
# every kernel on one stream, timed frames only: durations that add up to the frame
cd $root && tools/clean_trace.sh $tag > /dev/null || exit 1
cp $root/gpurun_out/clean_$tag/frame_split.md $out/frame_split_one_stream.md
cp $root/gpurun_out/clean_$tag/bench.json $out/bench_one_stream.json
ls -R $out | head -40
