"""Build the TEST-ONLY host simulation of the kernel bodies (tests/hostsim/hostsim.cpp)."""
import os
import subprocess
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
SRC = HERE / "hostsim.cpp"
EMU = HERE / "wave_emu_kernels.cpp"   # the wavefront-level kernels on emulated wavefronts (wave_emu.h)
DEPS = [SRC, EMU, HERE / "wave_emu.h"] + sorted((ROOT / "kalman-vae_amd" / "csrc").glob("*.h")) + [ROOT / "include" / "kvae_lgssm.h"]


def build(sanitize=False):
    out = HERE / ("libkvae_hostsim_asan.so" if sanitize else "libkvae_hostsim.so")
    if out.exists() and all(out.stat().st_mtime >= d.stat().st_mtime for d in DEPS):
        return out
    cmd = ["g++", "-std=c++17", "-shared", "-fPIC", "-Wall", "-Wextra", "-Wno-unused-parameter", "-pthread", "-o", str(out), str(SRC), str(EMU)]
    cmd += ["-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer"] if sanitize else ["-O2"]
    subprocess.run(cmd, check=True, cwd=ROOT)
    return out


if __name__ == "__main__":
    print(build(), build(sanitize=True))
