# host time per step of the DATA-PARALLEL step (one-rank RCCL group: every collective issued for real) replayed as hipGraph segments
# vs launched eagerly, for the headline and the three decoder workloads
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/hostms
for wl in mixed16384_neuradar mixed8192_vod_nll mixed16384_neuradar_full mixed16384_neuradar_full_fp16; do
  for seg in 1 0; do
    NR_SEGMENTS=$seg python3 bench.py --one-rank-collectives --workload $wl --steps 20 --warmup 20 --secondary= --full-model= --trained-steps 0 --no-cpu-baseline --no-roofline --no-render 2>gpurun_out/hostms/${wl}_$seg.err \
      | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['config']; print('$wl segments=$seg:', 'graph segments per step', c['graph_segments_per_step'], ' host', c['host_ms_per_step'], 'ms/step  GPU step', d['ms_per_step'], 'ms  (min', c['ms_per_step_min'], 'max', c['ms_per_step_max'], ') exchange', (c.get('main_table_exchange') or {}).get('mode'))" || tail -5 gpurun_out/hostms/${wl}_$seg.err
  done
done
