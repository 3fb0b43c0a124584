"""The short quotient the device box QPs use (ilqg_device.hpp div_plain: q = a * rd, q' = q + (a - b q) rd with rd the
correctly rounded 1 / b) against the C division on the host: equal bit for bit on random operands, including divisors
with extreme significands; differences only where the numerator is within 2^50 of the smallest normal number."""
import os
import subprocess

from conftest import ROOT


def test_short_quotient_equals_division(tmp_path):
    exe = str(tmp_path / "quotient_check")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tools", "ubench", "quotient_check.c"), "-lm"],
                   check=True)
    for lo, hi in ((-30, 30), (-300, 300), (-1, 0), (-200, 200)):
        out = subprocess.run([exe, "3000000", str(lo), str(hi)], check=True, capture_output=True, text=True).stdout
        assert out.strip().endswith(": 0 differ"), out
    # the documented limit: numerators below ~2^-998 (with divisors of the box QP's plain range) may differ
    out = subprocess.run([exe, "300000", "-1022", "-940", "-100", "100"], check=True, capture_output=True, text=True).stdout
    rows = [int(l.split(":")[1].split()[0]) for l in out.strip().splitlines()]
    assert rows[0] > 0 and all(r == 0 for r in rows[5:]), out
