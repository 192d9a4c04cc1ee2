#!/bin/bash
# GPU box: the round's closing evidence for ONE build of the kernels, in three calls (each well below gpurun's limit).
#   a: the -m gpu suite + rocprofv3 passes of the three scenes  (copy gpurun_out/prof_r05_*/summary/* into profiles/r05_*/ before b: bench.py
#      takes its roofline counts from the committed profile of the loaded kernels)
#   b: the three bench lines, every BASELINE config, the shard-scaling prediction, sweeps cover / cubes / book-one / cameras / scaled / wide
#   c: the general sweeps (depth 40, x4, depth 100)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
line() { python3 -c "
import json,sys
d=json.loads([l for l in open('gpurun_out/r05_bench_$1.log') if l.startswith(chr(123))][-1]); r=d['roofline']
print('$1', round(d['value'],1), round(d['ms_per_step'],2), round(r['kernel_ms'],2), round(d['single_render_ms'],2), r['frac'], r.get('valu_busy_frac_pmc'), r.get('reason'), d.get('image_matches_single_render'), d.get('oracle_pixels'), (d.get('cpu_baseline') or {}).get('value'))"; }
case "$1" in
a) python -m pytest tests -m gpu -q > gpurun_out/r05_gputest_final.log 2>&1; tail -3 gpurun_out/r05_gputest_final.log
   bash tools/profile_all.sh 2>&1 | grep -c "rc=0" ;;
b) python bench.py > gpurun_out/r05_bench_book_one.log 2>gpurun_out/r05_bench_book_one.err; line book_one
   python bench.py --scene cornell --width 600 --height 600 --spp 1000 > gpurun_out/r05_bench_cornell.log 2>&1; line cornell
   python bench.py --scene cover --width 800 --height 800 --spp 1000 > gpurun_out/r05_bench_cover.log 2>&1; line cover
   python tools/run_configs.py > gpurun_out/r05_run_configs.log 2>&1
   python tools/shard_scaling.py 2>/dev/null | tee gpurun_out/r05_shard_scaling.log
   python tools/shard_scaling.py --ascending 2>/dev/null | tee gpurun_out/r05_shard_scaling_ascending.log
   bash tools/r05_sweeps.sh 1; bash tools/r05_sweeps.sh 3 ;;
c) bash tools/r05_sweeps.sh 2; bash tools/r05_sweeps.sh 5; bash tools/r05_sweeps.sh 4 ;;
esac
