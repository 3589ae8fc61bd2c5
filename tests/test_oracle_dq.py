"""Pins oracle/dq_oracle.c against the reference's 21 DualQuaternionTest known answers
(test/quaternion_test.cpp, transcribed in tests/golden/dq_kat.json)."""
import json
import os

import numpy as np
import pytest

import oracle as O

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "dq_kat.json")))
TOL = KAT["tol"]
RAD = KAT["rad"]


def fix(name):
    a = KAT["fixtures"][name]
    ang = [RAD[x] for x in a[:3]]
    return O.dq_from_euler(ang[0], ang[1], ang[2], *[float(x) for x in a[3:]])


def ev(e):
    op = e[0]
    if op == "fix":
        return fix(e[1])
    if op == "rodrigues":
        return O.dq_from_rodrigues(e[1], [0, 0, 0])
    if op == "scale":
        return O.dq_scale(ev(e[1]), e[2])
    if op == "normalize":
        return O.dq_normalize(ev(e[1]))
    return {"add": O.dq_add, "sub": O.dq_sub, "mul": O.dq_mul}[op](ev(e[1]), ev(e[2]))


@pytest.mark.parametrize("case", KAT["cases"], ids=lambda c: c["name"])
def test_dq_known_answers(case):
    dq = ev(case["expr"])
    if case.get("expect_real") is not None:
        np.testing.assert_allclose(dq[:4], case["expect_real"], atol=TOL, rtol=0)
    if case.get("expect_real_of") is not None:
        np.testing.assert_allclose(dq[:4], ev(case["expect_real_of"])[:4], atol=TOL, rtol=0)
    if case.get("expect_dual") is not None:
        np.testing.assert_allclose(dq[4:], case["expect_dual"], atol=TOL, rtol=0)


@pytest.mark.parametrize("case", KAT["transforms"], ids=lambda c: c["name"])
def test_dq_transform_vertex(case):
    out = O.dq_transform_vertex(ev(case["expr"]), case["v"])
    np.testing.assert_allclose(out, case["expect"], atol=TOL, rtol=0)


def test_dq_compose_rotations():
    c = KAT["compose"]
    a, b = ev(c["a"]), ev(c["b"])
    twice = O.dq_transform_vertex(a, O.dq_transform_vertex(b, c["v"]))
    comp = O.dq_transform_vertex(O.dq_mul(a, b), c["v"])
    np.testing.assert_allclose(comp, twice, atol=TOL, rtol=0)


@pytest.mark.parametrize("case", KAT["angles"], ids=lambda c: c["name"])
def test_dq_angle_getters(case):
    get = {"roll": O.dq_roll, "pitch": O.dq_pitch, "yaw": O.dq_yaw}[case["getter"]]
    for name, expect in case["cases"]:
        assert abs(get(fix(name)) - RAD[expect]) <= TOL, (name, expect)


def test_dq_get_rodrigues():
    for name, expect in KAT["rodrigues_get"]["cases"]:
        np.testing.assert_allclose(O.dq_get_rodrigues(fix(name)), expect, atol=TOL, rtol=0)


def test_dq_to_string():
    # operator<< of boost quaternion: "(w,x,y,z)" with default ostream precision 6 (%g)
    t = KAT["tostring"]
    dq = fix(t["fix"])
    fmt = lambda q: "(" + ",".join("%g" % float(x) for x in q) + ")"
    assert "real: %s\ndual: %s\n" % (fmt(dq[:4]), fmt(dq[4:])) == t["expect"]


def test_dq_translation_roundtrip():
    dq = O.dq_from_euler(0.3, -0.2, 0.9, 1.5, -2.0, 0.25)
    np.testing.assert_allclose(O.dq_get_translation(dq), [1.5, -2.0, 0.25], atol=1e-5)


def test_half_roundtrip_all_values():
    # every finite half survives half->float->half; ties go to even
    for h in list(range(0, 0x7C00, 7)) + [0x7BFF, 0x0001, 0x03FF, 0x0400, 0x8001]:
        assert O.float_to_half(O.half_to_float(h)) == h
    assert O.float_to_half(np.float32(2.0 ** -25)) == 0  # tie to even (zero)
    assert O.float_to_half(np.float32(2.0 ** -25 * 1.0000001)) == 1
    assert O.float_to_half(1.0 + 2.0 ** -11) == 0x3C00  # tie -> even
    assert O.float_to_half(1.0 + 3 * 2.0 ** -11) == 0x3C02
    assert O.float_to_half(65520.0) == 0x7C00
    assert O.float_to_half(65519.0) == 0x7BFF
    # cross-check against numpy's IEEE half conversion on random floats
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(2000).astype(np.float32) * s for s in (1e-7, 1e-4, 1.0, 1e4)])
    ours = np.array([O.float_to_half(v) for v in x], np.uint16)
    assert np.array_equal(ours, x.astype(np.float16).view(np.uint16))
