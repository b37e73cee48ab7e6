"""CPU tier: the gfx950 shared library builds (hipcc cross-compiles without a GPU), loads, and exports EVERY entry point
that include/kvae_lgssm.h declares; the ctypes binding agrees with the header's ABI version.  No compute is called."""
import ctypes
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def declared_symbols():
    text = (ROOT / "include" / "kvae_lgssm.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kvae_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    sys.path.insert(0, str(ROOT))
    import __graft_entry__ as ge
    lib_path = ge.build_hip()
    dll = ctypes.CDLL(str(lib_path))
    names = declared_symbols()
    assert len(names) >= 20, names
    missing = [n for n in names if not hasattr(dll, n)]
    assert not missing, f"declared in include/kvae_lgssm.h but not exported: {missing}"
    from kvae import _native
    assert set(_native.SYMBOLS) == set(names), set(_native.SYMBOLS) ^ set(names)
    hdr = (ROOT / "include" / "kvae_lgssm.h").read_text()
    assert int(re.search(r"#define KVAE_ABI_VERSION (\d+)", hdr).group(1)) == _native.ABI_VERSION == dll.kvae_abi_version()


def test_hostsim_exports_the_same_abi():
    from hostsim.build import build
    dll = ctypes.CDLL(str(build()))
    missing = [n for n in declared_symbols() if not hasattr(dll, n)]
    assert not missing, missing
