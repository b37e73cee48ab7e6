# n = 4 forward/backward: one wavefront per sequence (lgssm_n4.h) vs sixteen sequences per wavefront (lgssm_m4.h)
for B in 256 1024 4096 32768; do
  echo "== wave-per-sequence B=$B"; KVAE_Q4=0 python3 tools/lgssm_probe.py --n 4 --B $B --T 50 --iters 10 2>&1 | grep -E "filter only|rts only|filter \+|backward"
  echo "== quad-per-sequence B=$B"; python3 tools/lgssm_probe.py --n 4 --B $B --T 50 --iters 10 --check 2>&1 | grep -E "filter only|rts only|filter \+|backward|oracle"
done
