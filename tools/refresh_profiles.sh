#!/bin/bash
# How profiles/ is produced (two GPU calls + one local step); TAG = r02, r03, ...
#
# 1. GPU box:   bash tools/refresh_profiles.sh gpu r02        (counters, kernel stats, probes, bench line)
# 2. locally:   bash tools/refresh_profiles.sh local r02      (ISA census + assemble profiles/)
# 3. GPU box again for the bench line once profiles/traffic.json has changed (bench.py reads it for
#    roofline.traffic / roofline.valu), then step 2 again.
set -u
mode=${1:-local}; tag=${2:-r02}
if [ "$mode" = gpu ]; then
  timeout 700 bash tools/pmc.sh pmc_$tag > /dev/null 2>&1
  timeout 250 tools/kstats.sh ks_${tag}_if1 --inflight 1 --steps 20 --no-extras --sustain-seconds 0 > /dev/null
  timeout 250 tools/kstats.sh ks_${tag}_default --no-extras --sustain-seconds 0 > /dev/null
  timeout 120 tools/bin/valu_rate_probe > gpurun_out/valu_rates_$tag.txt 2>&1
  timeout 120 tools/bin/chain_probe > gpurun_out/chain_probe_$tag.txt 2>&1
  timeout 900 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
  timeout 200 python bench.py --mode fast32 --cpu-fields 0 --no-extras > gpurun_out/bench_${tag}_fast32.json 2>> gpurun_out/bench_$tag.err
  # the command the driver runs at round end, exactly (its own K / W, every side leg): the contract line and what it left in
  # bench_extras.json
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_${tag}_driver_cmd.json 2>> gpurun_out/bench_$tag.err
  cp bench_extras.json gpurun_out/bench_extras_$tag.json 2> /dev/null
  # NTSCSIM_MODE_FLOAT (round 6): bench line, kernel stats, error against the oracle, stall counters of its forms, HBM bytes
  timeout 200 python bench.py --mode float --cpu-fields 0 --no-extras > gpurun_out/bench_${tag}_float.json 2>> gpurun_out/bench_$tag.err
  timeout 250 tools/kstats.sh ks_${tag}_float --mode float --inflight 1 --steps 20 --no-extras --sustain-seconds 0 > /dev/null
  timeout 600 python tools/float_err.py --mode float > gpurun_out/float_err_$tag.txt 2>&1
  timeout 400 bash tools/fp_probe.sh fp_$tag "10 0" "10 0" > gpurun_out/float_pmc_$tag.txt 2>&1
  ( export TMPDIR=/tmp; R=$PWD; cd /tmp; for grp in FETCH_SIZE WRITE_SIZE; do
      timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/fp_$tag/$grp -o pmc -- python $R/bench.py --mode float --steps 2 --warmup 1 --cpu-fields 0 --inflight 1 --no-extras --sustain-seconds 0 > $R/gpurun_out/fp_$tag/$grp.log 2>&1 < /dev/null; done )
  python tools/pmc_summary.py gpurun_out/fp_$tag > gpurun_out/float_traffic_$tag.txt 2>&1
  # the synchronous call: shipped chain, two-launch form, the three roles side by side (ceiling of a pipelined form), float
  timeout 300 python tools/role_probe.py > gpurun_out/role_probe_$tag.txt 2>&1
  # ... and the form that ships: the call as a pipeline of wavefront roles (rates, kernel trace, role clocks, submit by depth)
  timeout 900 tools/sync_trace.sh $tag > /dev/null 2>&1             # -> gpurun_out/sync_$tag.txt
  timeout 300 bash tools/sync422_trace.sh > /dev/null 2>&1          # -> gpurun_out/sync422_timeline.txt
  timeout 500 bash tools/pmc422.sh pmc422_$tag > /dev/null 2>&1
  timeout 250 tools/kstats.sh ks_${tag}_tocomp --tool to_composite --inflight 1 --steps 20 --sustain-seconds 0 > /dev/null
  timeout 600 python bench.py --tool to_composite --cpu-fields 200 > gpurun_out/bench_${tag}_tocomp.json 2>> gpurun_out/bench_$tag.err
  ( export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/ks_${tag}_raw28; mkdir -p $O; cd /tmp;
    timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o ks -- python $R/tools/raw28_probe.py > $O/probe.log 2>&1 < /dev/null;
    f=$(find $O -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats.csv )
  timeout 300 bash tools/pmc_raw28.sh gpurun_out/raw28_front_pmc_$tag.txt > /dev/null 2>&1
  timeout 60 tools/bin/follow_probe > gpurun_out/follow_probe_$tag.txt 2>&1
  timeout 200 bash tools/fetch_calibrate.sh > /dev/null 2>&1       # -> gpurun_out/fetch_calibration.txt
  timeout 300 sh tools/submit_probe.sh > /dev/null 2>&1            # -> gpurun_out/submit_probe.txt
  timeout 120 bash tools/dryrun_two_ranks_one_gpu.sh nccl > /dev/null 2>&1   # -> gpurun_out/dryrun_two_ranks.txt
  timeout 400 sh tools/host422_loop_probe.sh > /dev/null 2>&1      # -> gpurun_out/host422_loop_probe.txt  (round 5)
  # the synchronous one-field call under the kernel trace: where its 0.54 ms go (round 5)
  ( export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/ks_${tag}_sync; mkdir -p $O; cd /tmp;
    timeout 120 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O -o ks -- $R/composite-video-simulator_amd/field_loop -vhs --mode sync --fields 300 --warmup 50 > $O/probe.log 2>&1 < /dev/null;
    f=$(find $O -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats.csv;
    f=$(find $O -name "*memory_copy_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/memory_copy_stats.csv )
  timeout 120 composite-video-simulator_amd/rank_bench -vhs --spawn 1 --frames 300 --steps 40 --warmup 8 > gpurun_out/rank_bench_$tag.txt 2>&1
  tail -c 600 gpurun_out/bench_$tag.json; tail -c 400 gpurun_out/bench_${tag}_tocomp.json
else
  S=composite-video-simulator_amd/csrc
  mkdir -p /tmp/census_$tag
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-sched-strategy=iterative-maxocc -Iinclude -I$S --offload-arch=gfx950 -S --cuda-device-only \
      -o /tmp/census_$tag/ntscsim.s $S/ntscsim_hip.hip 2> /dev/null
  for k in 'k_decode_fastILb1EdLb0EE:4' 'k_decode_fastILb0EdLb0EE:4' 'k_encode_fastIdE:16' 'k_row_states:1' 'k_field_setup:1'; do
    python tools/isa_cost.py /tmp/census_$tag/ntscsim.s "${k%%:*}" --steps "${k##*:}" --json "/tmp/census_$tag/${k%%:*}.json" | head -1
  done
  (cd tools && python loop_census.py /tmp/census_$tag/ntscsim.s 'k422_fusedILb1ELb1E' --mean > /tmp/census_$tag/k422_fused.json)
  [ -s gpurun_out/fetch_calibration.txt ] && cp gpurun_out/fetch_calibration.txt profiles/r04_fetch_calibration.txt
  # (since round 6 bench.py prints the contract object only: the full dictionary make_profiles.py reads is the driver-command
  #  run's bench_extras.json; the printed lines are kept beside it)
  [ -s gpurun_out/bench_$tag.json ] && cp gpurun_out/bench_$tag.json profiles/${tag}_bench_line_default_cmd.json
  python tools/make_profiles.py $tag gpurun_out/bench_extras_$tag.json gpurun_out/ks_${tag}_default gpurun_out/ks_${tag}_if1 \
      gpurun_out/pmc_$tag gpurun_out/valu_rates_$tag.txt gpurun_out/chain_probe_$tag.txt /tmp/census_$tag \
      gpurun_out/bench_${tag}_fast32.json gpurun_out/bench_${tag}_tocomp.json gpurun_out/ks_${tag}_tocomp \
      gpurun_out/pmc422_$tag > /dev/null && echo "profiles/ assembled"
  [ -s gpurun_out/ks_${tag}_raw28/kernel_stats.csv ] && cp gpurun_out/ks_${tag}_raw28/kernel_stats.csv profiles/${tag}_kernel_stats_raw28.csv
  [ -s gpurun_out/raw28_front_pmc_$tag.txt ] && { cat gpurun_out/raw28_front_pmc_$tag.txt; echo; echo "tools/follow_probe.hip (one follower step of a lone wavefront, from registers):"; grep -E "wave\(s\)|workgroup" gpurun_out/follow_probe_$tag.txt; } > profiles/${tag}_raw28_front_pmc.txt
  [ -s gpurun_out/bench_${tag}_driver_cmd.json ] && cp gpurun_out/bench_${tag}_driver_cmd.json profiles/${tag}_bench_driver_cmd.json
  [ -s gpurun_out/submit_probe.txt ] && cp gpurun_out/submit_probe.txt profiles/${tag}_submit_probe.txt
  [ -s gpurun_out/dryrun_two_ranks.txt ] && cp gpurun_out/dryrun_two_ranks.txt profiles/${tag}_dryrun_two_ranks.txt
  [ -s gpurun_out/host422_loop_probe.txt ] && cp gpurun_out/host422_loop_probe.txt profiles/${tag}_host422_loop_probe.txt
  [ -s gpurun_out/ks_${tag}_sync/kernel_stats.csv ] && { cat gpurun_out/ks_${tag}_sync/kernel_stats.csv; echo; cat gpurun_out/ks_${tag}_sync/memory_copy_stats.csv; } > profiles/${tag}_sync_call_stats.csv
  [ -s gpurun_out/rank_bench_$tag.txt ] && grep '^{' gpurun_out/rank_bench_$tag.txt > profiles/${tag}_rank_bench.json
  [ -s gpurun_out/bench_extras_$tag.json ] && cp gpurun_out/bench_extras_$tag.json profiles/${tag}_bench_extras.json
  [ -s gpurun_out/bench_${tag}_float.json ] && cp gpurun_out/bench_${tag}_float.json profiles/${tag}_bench_float.json
  [ -s gpurun_out/ks_${tag}_float/kernel_stats.csv ] && cp gpurun_out/ks_${tag}_float/kernel_stats.csv profiles/${tag}_kernel_stats_float.csv
  [ -s gpurun_out/float_err_$tag.txt ] && cp gpurun_out/float_err_$tag.txt profiles/${tag}_float_err.txt
  [ -s gpurun_out/float_pmc_$tag.txt ] && { cat gpurun_out/float_pmc_$tag.txt; echo; echo "== FETCH_SIZE / WRITE_SIZE (KiB, raw counters; calibration factors: profiles/r04_fetch_calibration.txt)"; cat gpurun_out/float_traffic_$tag.txt; } > profiles/${tag}_float_pmc.txt
  [ -s gpurun_out/role_probe_$tag.txt ] && cp gpurun_out/role_probe_$tag.txt profiles/${tag}_role_probe.txt
  [ -s gpurun_out/sync_$tag.txt ] && cp gpurun_out/sync_$tag.txt profiles/${tag}_sync_pipe.txt
  [ -s gpurun_out/sync422_timeline.txt ] && cp gpurun_out/sync422_timeline.txt profiles/${tag}_sync422_timeline.txt
  # opcode histograms of the hand-tuned decoder forms' steady loops (the per-stage census: profiles/${tag}_decode_census.txt)
  { for k in 'k_decode_fastILb1EdLb0EE' 'k_decode_fast_xiIdE' 'k_decode_fast_foIdE' 'k_decode_fast_svIdE'; do
      echo "== $k"; (cd tools && python loop_census.py /tmp/census_$tag/ntscsim.s "$k" --hist | awk 'NR % 2 == 1 || 1' | cut -c1-1400 | grep -A1 "VALU [67][0-9][0-9] " | head -2); done; } > profiles/${tag}_loop_histograms.txt 2>/dev/null
fi
