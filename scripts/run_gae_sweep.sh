mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -30 > gpurun_out/test6.log
for L in 8 16; do for EB in 64 32 16; do CRL_GAE_L=$L CRL_GAE_EB=$EB timeout 120 python scripts/bench_gae.py >> gpurun_out/gae_sweep.txt 2>&1; done; done
timeout 120 python scripts/bench_gae.py >> gpurun_out/gae_sweep.txt 2>&1
echo done
