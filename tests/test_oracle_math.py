"""The build's own natural-log spec (oracle/orc_math.h: orc_logf; the product states the same in csrc/nrc_math.h): the reference calls GLSL
`log` (data/shader/include/path_trace.glsl:36,163), whose result is implementation defined within Vulkan's precision contract -- 3 ulp outside
[0.5, 2], absolute error < 2^-21 inside.  The integrator can only pass 1 - u with u = k * 2^-23, so the whole domain is checked, every point."""
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def domain_and_log(orc):
    x = (np.arange(1, 2 ** 23 + 1, dtype=np.float64) * 2.0 ** -23).astype(np.float32)      # every 1 - k * 2^-23, k = 2^23 - 1 .. 0
    return x, orc.math_eval(0, x)[0]


def test_log_spec_meets_the_glsl_precision_contract_on_every_argument_the_integrator_can_pass(domain_and_log):
    x, y = domain_and_log
    ref = np.log(x.astype(np.float64))
    ref32 = ref.astype(np.float32)
    ulp = np.spacing(np.abs(ref32)).astype(np.float64)
    ulp[ulp == 0] = 2.0 ** -149
    err = np.abs(y.astype(np.float64) - ref) / ulp
    assert err.max() < 1.75, err.max()                                  # measured 1.68; GLSL: 3 ulp outside [0.5, 2]
    inside = x >= 0.5
    assert np.abs(y[inside] - ref[inside]).max() < 2.0 ** -23           # GLSL: 2^-21 inside [0.5, 2]
    assert (y == ref32).mean() > 0.7                                    # correctly rounded for three arguments in four
    assert y[-1] == 0.0 and np.signbit(y[-1]) == False                  # log(1) == +0: a zero-length free flight for u == 0
    assert y[-2] == np.float32(np.log(1.0 - 2.0 ** -23))                # no cancellation just below 1
    assert (np.diff(y) >= 0).all()                                      # monotone over the whole domain
    assert (y[:-1] < 0).all()


def parse_inc(path):
    text = open(path).read()
    rows = re.findall(r"\{\s*(-?0x[0-9a-f.]+p[+-]?\d+)f,\s*(-?0x[0-9a-f.]+p[+-]?\d+)f\}", text)
    return [(float.fromhex(a), float.fromhex(b)) for a, b in rows]


def test_log_tables_of_product_and_oracle_are_the_exactly_rounded_table():
    """tools/make_log_table.py derives the table with exact arithmetic; the product's copy and the oracle's copy hold those numbers"""
    import make_log_table
    want = make_log_table.table()
    assert len(want) == 128 and want[0] == (2.0, -float(np.float32(np.log(2.0)))) and want[127] == (1.0, 0.0)
    for path in ("nrc-hpm-renderer_amd/csrc/nrc_log_table.inc", "oracle/orc_log_table.inc"):
        assert parse_inc(os.path.join(ROOT, path)) == want, path
    ln2, third = make_log_table.constants()
    for path, names in (("nrc-hpm-renderer_amd/csrc/nrc_math.h", ("NRC_LN2", "NRC_THIRD")), ("oracle/orc_math.h", ("ORC_LN2", "ORC_THIRD"))):
        text = open(os.path.join(ROOT, path)).read()
        got = [float.fromhex(re.search(r"#define %s (\S+)f" % n, text).group(1)) for n in names]
        assert got == [ln2, third], (path, got)
    # every inv_c is the float nearest 1 / centre and every log_c the float nearest -ln(inv_c): spot-check against double precision
    for i, (inv_c, log_c) in enumerate(want):
        c = 0.5 if i == 0 else 1.0 if i == 127 else 0.5 + (2 * i + 1) / 512.0
        assert inv_c == float(np.float32(1.0 / c))
        assert abs(log_c + np.log(inv_c)) <= np.spacing(np.float32(abs(log_c))) * 0.51
