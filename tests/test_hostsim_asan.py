"""Run a slice of the host-simulated kernels under AddressSanitizer + UBSan (GPU sanitizers are unavailable on the
pool, so the CPU build of the kernel bodies is the sanitizer target).  Executed in a subprocess with libasan
preloaded; any heap/stack overflow or UB in the bodies aborts it."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]

SCRIPT = r"""
import sys
sys.path[:0] = [%(root)r, %(root)r + "/kalman-vae_amd", %(root)r + "/tests"]
import torch
from kvae import _native
from hostsim.build import build
_native._set_test_backend(_native.LgssmLib(build(sanitize=True)))
import parity_cases
for dims in ((3, 9, 4, 4, 2, 3), (2, 7, 3, 2, 1, 2), (2, 5, 16, 16, 2, 1), (1, 1, 4, 4, 2, 3)):
    parity_cases.vs_oracle_random("cpu", *dims)
parity_cases.safe_cholesky_levels("cpu")
parity_cases.lstm_vs_torch("cpu", 2, 5, 2, 50)
parity_cases.regime_vs_torch("cpu", 2, 6, 3, 0.7, False)
parity_cases.vae_epilogue_vs_torch("cpu", 2, 3, 4, 4, 2, True)
parity_cases.conv_edge_vs_torch("cpu", 2)
parity_cases.enc_mid_vs_torch("cpu", 2, 8)
parity_cases.dec_up_vs_torch("cpu", 2, 4)
parity_cases.vae_heads_vs_torch("cpu", 5)
# the product's wavefront-level kernels (lgssm_m4.h / lgssm_q4.h, lgssm_n16.h) on emulated wavefronts (hostsim/wave_emu.h):
# ragged last wavefront (19 sequences, 16 per wavefront), the tails of the unrolled loops (T = 3, T = 1), 16-byte accesses
lib = _native.lib_for(torch.zeros(1))
lib.dll.kvae_hostsim_wave_emu(1)
for dims in ((19, 3, 4, 4, 2, 3), (1, 1, 4, 4, 2, 2), (2, 3, 16, 16, 2, 2)):
    parity_cases.vs_oracle_random("cpu", *dims)
lib.dll.kvae_wemu_m4_split_max_b(0)      # (4,4,2) again in its single-launch form (the one batches above 2048 sequences take)
parity_cases.vs_oracle_random("cpu", 19, 4, 4, 4, 2, 3)
lib.dll.kvae_wemu_m4_split_max_b(-1)
assert min(lib.dll.kvae_wemu_launches(i) for i in range(4)) > 0
lib.dll.kvae_hostsim_wave_emu(0)
print("ASAN-OK")
"""


def test_hostsim_under_asan():
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not asan or not Path(asan).exists():
        pytest.skip("libasan not available")
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1",
               OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": str(ROOT)}], env=env, capture_output=True, text=True,
                       timeout=900)
    assert "ASAN-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
