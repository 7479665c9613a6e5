"""GPU: the C ABI used from a plain C program (tests/cabi/score_demo.c) - no Python or torch in the process."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("prog", ["score_demo", "train_demo"])
def test_plain_c_program_through_the_c_abi(vsa, tmp_path, prog):
    """score_demo.c: scoring (include/vs_scorer.h); train_demo.c: one training step (include/vs_train.h)."""
    gcc = shutil.which("gcc")
    assert gcc, "gcc not found"
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    pkg = os.path.join(ROOT, "video-summarization_amd")
    exe = str(tmp_path / prog)
    build = subprocess.run([gcc, "-std=gnu99", "-O2", os.path.join(ROOT, "tests", "cabi", prog + ".c"),
                            "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(rocm, "include"), "-D__HIP_PLATFORM_AMD__",
                            "-L" + pkg, "-lvsscore", "-L" + os.path.join(rocm, "lib"), "-lamdhip64",
                            "-Wl,-rpath," + pkg, "-Wl,-rpath," + os.path.join(rocm, "lib"), "-lm", "-o", exe],
                           capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and run.stdout.startswith("OK"), (run.stdout, run.stderr[-2000:])
