"""CPU: the parity contract itself (tests/parity.py) — which bound a flag combination may claim, and that nothing but a
flag excuses a farm.  The contract is what every GPU test and every fuzzer asserts, so its own logic is pinned here."""
import numpy as np

import parity


def _e(**kw):
    base = dict(power=0.0, ws=0.0, wd=0.0, ti=0.0, std=0.0)
    base.update(kw)
    return {k: np.array([v], dtype=np.float64) for k, v in base.items()}


def test_unflagged_farms_get_the_plain_tolerances():
    assert parity.within(_e(power=9e-5, ws=4e-5, wd=2.9e-4, ti=4e-6, std=9e-5), parity.TOL).all()
    for k, v in parity.TOL.items():
        assert not parity.within(_e(**{k: v * 1.01}), parity.TOL).any(), k
    # farms of more than 128 turbines: three times the tolerance on power / speed / std, not on direction or TI
    assert parity.within(_e(power=2.9e-4), parity.TOL, n_turbines=200).all()
    assert not parity.within(_e(wd=3.1e-4), parity.TOL, n_turbines=200).any()


def test_a_flag_only_excuses_what_its_event_can_move():
    knee, ramp, ovl = parity.RISK_POWER_KNEE, parity.RISK_THRUST_RAMP, parity.RISK_OVERLAP
    # power knee alone: the wind field must be as good as an unflagged farm's, only the power may move
    assert parity.flagged_within(_e(power=4e-2), np.array([knee]), 80).all()
    assert not parity.flagged_within(_e(power=4e-2, wd=4e-4), np.array([knee]), 80).any()
    assert not parity.flagged_within(_e(power=6e-2), np.array([knee]), 80).any()
    # thrust ramp without an overlap flag: a few TOL
    assert parity.flagged_within(_e(power=9e-3, ws=9e-4, wd=9e-3), np.array([ramp]), 80).all()
    assert not parity.flagged_within(_e(wd=2e-2), np.array([ramp]), 80).any()
    assert parity.flagged_within(_e(power=9e-3, ws=9e-4), np.array([ramp | knee]), 80).all()
    # overlap flip: the bounded signature of one count flipping
    assert parity.flagged_within(_e(power=9e-2, ws=1.9e-2, wd=9e-2, ti=1.9e-2), np.array([ovl]), 80).all()
    assert not parity.flagged_within(_e(power=0.2), np.array([ovl]), 80).any()
    assert not parity.flagged_within(_e(wd=0.15), np.array([ovl | knee]), 80).any()
    # overlap flip at a turbine on the thrust ramp: the two amplifiers compound (fuzz_api seed 501, session 64)
    assert parity.flagged_within(_e(power=0.27, ws=2.7e-2, wd=0.13), np.array([ovl | ramp]), 42).all()
    assert not parity.flagged_within(_e(power=0.45), np.array([ovl | ramp]), 42).any()  # 1.5x the measurement, no more
    assert not parity.flagged_within(_e(wd=0.25), np.array([ovl | ramp]), 42).any()
    # farms of more than 128 turbines: the knee / ramp bounds widen on power / speed / std only, like TOL
    assert parity.flagged_within(_e(power=0.12), np.array([knee]), 200).all()
    assert not parity.flagged_within(_e(wd=4e-4), np.array([knee]), 200).any()
    assert not parity.flagged_within(_e(wd=2e-2), np.array([ramp]), 200).any()
    assert not parity.flagged_within(_e(ti=3e-4), np.array([ramp]), 200).any()
    assert not parity.flagged_within(_e(ti=3e-2), np.array([ovl | ramp | knee]), 42).any()
    # the cut-out drop of the power table (5 MW -> 0 within 0.01 m/s): a knee-flagged farm is judged against RATED power there
    drop = _e(power=6.2e-2)
    drop["power_of_rated"] = np.array([4e-3])
    assert parity.flagged_within(drop, np.array([knee]), 91).all() and parity.flagged_within(drop, np.array([knee | ramp]), 91).all()
    drop["power_of_rated"] = np.array([3e-2])
    assert not parity.flagged_within(drop, np.array([knee]), 91).any()
    assert not parity.flagged_within(drop, np.array([ramp]), 91).any()  # (no knee flag: no such allowance)


def test_check_strict_admits_no_exemption():
    B, N = 4, 5
    ref = dict(power=np.full((B, N), 2.0e6), wind_speed=np.full((B, N), 8.0), wind_direction=np.full((B, N), 270.0),
               load=np.full((B, N, 4), 0.06))
    got = {k: v.astype(np.float32) for k, v in ref.items()}
    parity.check_strict(got, ref)
    got["power"][2, 3] *= 1.0 + 2e-4
    try:
        parity.check_strict(got, ref)
    except AssertionError as e:
        assert "float64 re-solve" in str(e) and "[2]" in str(e)
    else:
        raise AssertionError("a farm 2e-4 off in power passed the strict check")
