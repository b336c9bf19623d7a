for n in 32768 65536; do for rep in 1 2; do for lib in nuclear_sim_amd/libnpb.so nuclear_sim_amd/ablate/libnpb_sleep0.so nuclear_sim_amd/ablate/libnpb_sleep3.so; do
NPB_LIB=$lib python3 bench.py --plants-per-gpu $n --steps 400 --warmup 50 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$n', '$lib', d['ms_per_step'], d['roofline']['frac'])"
done; done; done
for seg in 8192 12288 16384 20480 24576; do for rep in 1 2; do
NPB_ARENA_SEGMENT=$seg python3 bench.py --plants-per-gpu 65536 --steps 400 --warmup 50 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('65536 segment $seg', d['ms_per_step'], d['roofline']['frac'])"
done; done
