"""The C ABI from plain C: tests/c_abi/abi_parity.c (gcc, no Python/torch in the process) drives sampling → renumbering →
feature gather through libwholegraph_amd.so with the default allocator callbacks and checks every output bit-for-bit
against the C oracle.  This is the drop-in boundary as a libwholegraph C/C++ client sees it."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CDIR = os.path.join(ROOT, "tests", "c_abi")


def _build():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s"], check=True)
    subprocess.run(["make", "-C", CDIR, "-s"], check=True)
    return os.path.join(CDIR, "build", "abi_parity")


def test_c_client_compiles_and_links_without_hipcc():
    exe = _build()
    assert os.access(exe, os.X_OK)
    out = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libwholegraph_amd.so" in out and "libtorch" not in out and "libpython" not in out


@pytest.mark.gpu
def test_c_client_parity_on_gpu():
    exe = _build()
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "C_ABI_PARITY_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]
    assert "invalid input" in p.stderr or "logic error" in p.stderr   # the deliberate wrong-dtype call logged one line
