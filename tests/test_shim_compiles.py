"""The reference's own drivers must compile UNCHANGED against include/FHEController.h (drop-in boundary,
SURVEY.md §8(b)).  Reads /root/reference as text, CPU container only (the GPU box has no reference tree)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"


@pytest.mark.parametrize("driver", ["main.cpp", "main_2.cpp"])
def test_reference_driver_compiles_against_shim(driver):
    path = os.path.join(REF, driver)
    if not os.path.exists(path):
        pytest.skip("reference tree not present")
    # fed through stdin so that `#include "FHEController.h"` resolves to OUR include/ directory
    with open(path, "rb") as src:
        r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), "-x", "c++", "-"],
                           stdin=src, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]


def test_shim_driver_builds():
    out = os.path.join(ROOT, "tests", "shim", "shim_driver")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "shim", "shim_driver.cpp"),
                        "-L", os.path.join(ROOT, "fhe-linformer_amd"), "-lfhelin_amd", "-Wl,-rpath," + os.path.join(ROOT, "fhe-linformer_amd"),
                        "-o", out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    assert os.path.exists(out)
