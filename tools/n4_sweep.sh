for B in 256 1024 4096 32768; do
  echo "== wave-per-sequence B=$B"; python3 tools/lgssm_probe.py --n 4 --B $B --T 50 --iters 10 2>&1 | grep -E "filter \+|elbo|backward"
  echo "== thread-per-sequence B=$B"; KVAE_N4_TPP_FWD_MIN_B=1 python3 tools/lgssm_probe.py --n 4 --B $B --T 50 --iters 10 --check 2>&1 | grep -E "filter \+|backward|oracle mus_smooth|oracle Sigmas_smooth"
done
