cd $GRAFT_REPO_ROOT/sdr_receiver_dvb_t2_amd && cp libt2gpu.so /tmp/new.so
run() { python $GRAFT_REPO_ROOT/bench.py --config 4 --no-cpu-baseline --no-extra-legs --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['host_end']['ts_bytes_per_s'], [(k['stage'][:5],k['ms']) for k in d['roofline']['kernels']])"; }
cp libt2gpu_head.so.keep libt2gpu.so; run head
cp /tmp/new.so libt2gpu.so; run new
cp libt2gpu_head.so.keep libt2gpu.so; run head
cp /tmp/new.so libt2gpu.so; run new
