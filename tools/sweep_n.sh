for n in 40960 49152 53248 57344 59392 61440 63488 65536 69632 73728; do echo "n=$n"; NPB_AB_N=$n python3 tools/ab_kernel.py nuclear_sim_amd/libnpb.so | head -1; done
