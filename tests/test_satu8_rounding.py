"""SatU8 of the augmentation kernels (dali_amd/csrc/augment.hip) is ConvertSat<uint8_t> in three instructions; the C program
proves the identity for every float in [-1000, 1000] (2.3 G values, about 7 s)."""
import os
import subprocess


def test_one_addition_rounds_half_away_for_every_float(tmp_path):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "satu8_check")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(root, "tools", "satu8_check.c"), "-lm"])
    out = subprocess.check_output([exe], timeout=300).decode()
    assert out.strip().endswith(" 0 mismatches"), out
