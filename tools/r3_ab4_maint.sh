for n in 32768 65536; do for rep in 1 2; do for lib in nuclear_sim_amd/ablate/libnpb_prev.so nuclear_sim_amd/libnpb.so; do
NPB_LIB=$lib python3 bench.py --plants-per-gpu $n --steps 400 --warmup 50 --no-cpu-baseline --maintenance 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$n maintenance', '$lib', d['ms_per_step'], d['roofline']['frac'])"
done; done; done
