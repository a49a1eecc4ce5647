"""The C++ host layer (flac_amd/host/flac.hpp: the reference's Go types and Read() restated over the C ABI)
against the expectations of the reference's own tests, flacmeta_test.go:10-278.  The test program is built
in-tree by __graft_entry__.build() / make -C tests/host_cpp."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
EXE = os.path.join(HERE, "host_cpp", "test_flac_host")


def _exe():
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-C", os.path.join(HERE, "host_cpp"), "-s"])
    return EXE


def test_metadata_like_the_reference_tests(testdata):
    out = subprocess.run([_exe(), testdata], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all checks passed" in out.stdout


@pytest.mark.gpu
def test_decode_fixtures_through_cpp_host(testdata):
    out = subprocess.run([_exe(), testdata, "--gpu"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "with GPU decode" in out.stdout
