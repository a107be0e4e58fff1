"""-m gpu: the C ABI used from plain C (tests/c_client/c_client.c, gcc, no Python in the loop):
robot and obstacle handed over through the header's structs, the reference's command strings
through orc_send_command, a batch through the kernel-level entry points, error messages."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_is_plain_c(tmp_path):
    """include/orcdchomp_amd.h compiles as C99 with -pedantic (no GPU needed)"""
    src = tmp_path / "hdr.c"
    src.write_text('#include "orcdchomp_amd.h"\nint main(void) { return 0; }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                    "-fsyntax-only", str(src)], check=True)


@pytest.mark.gpu
def test_c_client(tmp_path):
    lib_dir = os.path.join(ROOT, "or_cdchomp_amd")
    exe = tmp_path / "c_client"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c_client", "c_client.c"), "-o", str(exe),
                    "-L", lib_dir, "-lorcdchomp_amd", "-lm", "-Wl,-rpath," + lib_dir], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    sys.stdout.write(out.stdout)
    sys.stderr.write(out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "C CLIENT OK" in out.stdout
