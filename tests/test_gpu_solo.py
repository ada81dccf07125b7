"""The solo form of an evaluation (frx_solo_kernel.hpp): one workgroup per candidate runs forward map, penalty integral and adjoint in ONE launch - the form of
batches larger than the one-launch clusters reach (Monte-Carlo scale, BASELINE configs[3..4]) and of the per-stage rounds of frx_optimize on such batches.
It calls the stage kernels' own bodies on the same stage buffers, so every result has to equal the three launches' BIT FOR BIT; the oracle comparison
(se3gcopter_cpu.hpp:961-1000 restated, 1e-9) is made directly as well."""
import numpy as np
import pytest

PER_EVAL_TOL = 1e-9


def three_launches(prob, x):
    prob.set_eval_solo(0)
    fused = prob.eval_fused()
    prob.set_eval_fused(False)
    f, g = prob.objective(x)
    if fused: prob.set_eval_fused(True)
    return f, g


@pytest.mark.gpu
@pytest.mark.parametrize("kappa", [16, 48, 8, 63])
def test_solo_equals_three_launches_bit_for_bit_and_the_oracle(frx, sc, ob, kappa):
    """Ragged batch (2 .. 64 pieces, with and without obstacles; K_i 8 .. 14): the instantiations for kappa = 16 / 48 and the run-time one (kappa = 8: 28 pieces per
    pass; kappa = 63: four pieces per pass, 16 passes for 64 pieces)."""
    cands = [sc.make_candidate(60, 64, 16, obstacles=True), sc.make_candidate(61, 33, 8), sc.make_candidate(62, 2, 0), sc.make_candidate(63, 17, 4, obstacles=True),
             sc.make_candidate(64, 64, 16, perturb_id=3), sc.make_candidate(65, 5, 1), sc.make_candidate(66, 48, 12, obstacles=True)]
    prob = frx.Problem(cands, sc.ZHANGJIAJIE, qd_intervals=kappa)
    oracles = [ob.Oracle(c, sc.ZHANGJIAJIE, qd_intervals=kappa) for c in cands]
    for o in oracles: o.set_abscissa_mode(False)
    x0 = prob.initial_guess()
    prob.set_resident(False)
    xs = [x0, prob.optimize(1e-6, x0=x0, max_iterations=25)["x"]]
    for s, x in enumerate(xs):
        f3, g3 = three_launches(prob, x)
        prob.set_eval_solo(2)
        assert prob.eval_solo() >= 1
        f1, g1 = prob.objective(x)
        assert np.array_equal(f1, f3), f"state {s}: objective values differ"
        assert np.array_equal(g1, g3), f"state {s}: gradients differ"
        for b, o in enumerate(oracles):
            sl = slice(prob.x_off[b], prob.x_off[b + 1])
            f_ref, g_ref = o.objective(x[sl])
            assert abs(f1[b] - f_ref) <= PER_EVAL_TOL * abs(f_ref), f"state {s} cand {b}"
            assert np.abs(g1[sl] - g_ref).max() <= PER_EVAL_TOL * max(np.abs(g_ref).max(), abs(f_ref)), f"state {s} cand {b}"
    prob.close()


@pytest.mark.gpu
def test_solo_is_the_default_from_its_threshold_on_and_replays_in_a_graph(frx, sc):
    """448 candidates (between BASELINE configs[3]'s batch and a GPU's share of configs[4]): the default evaluation is the solo launch, identical to the three launches; the capturable
    _device form replays in a hipGraph (nothing on the host between two of them) - in a process of its own that loads torch before the library, as bench.py does."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import json, sys
sys.path.insert(0, %r)
import numpy as np, torch
from frx_import import frx
from fast_racing_amd import scenario as sc
B0, N, gates, kappa = sc.CONFIGS["headline"]
prob = frx.Problem([sc.make_candidate(0, N, gates, perturb_id=b) for b in range(448)], sc.ZHANGJIAJIE, qd_intervals=kappa)
default_solo, default_fused = prob.eval_solo(), prob.eval_fused()
x = prob.initial_guess()
f1, g1 = prob.objective(x)
prob.set_eval_solo(0)
f3, g3 = prob.objective(x)
prob.set_eval_solo(1)
xd = torch.from_numpy(x).cuda(); fd = torch.zeros(prob.B, dtype=torch.float64, device="cuda"); gd = torch.zeros(prob.NX, dtype=torch.float64, device="cuda")
s = torch.cuda.Stream(); same = []
with torch.cuda.stream(s):
    prob.objective_device(xd.data_ptr(), fd.data_ptr(), gd.data_ptr(), s.cuda_stream); s.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s):
        for _ in range(5): prob.objective_device(xd.data_ptr(), fd.data_ptr(), gd.data_ptr(), torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        fd.zero_(); gd.zero_(); gr.replay(); s.synchronize()
        same.append(bool(np.array_equal(fd.cpu().numpy(), f3) and np.array_equal(gd.cpu().numpy(), g3)))
print(json.dumps({"default_solo": default_solo, "default_fused": default_fused, "blocking_same": bool(np.array_equal(f1, f3) and np.array_equal(g1, g3)), "replays_same": same}))
""" % root
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["default_solo"] >= 1 and r["default_fused"] == 0 and r["blocking_same"] and r["replays_same"] == [True, True, True], r


@pytest.mark.gpu
def test_per_stage_rounds_on_the_solo_launch_give_the_same_plan(frx, sc):
    """frx_optimize on the per-stage path (line-search tap, skipped candidates, completion count in the adjoint's epilogue): with the solo launch every round is
    TWO launches (k_lbfgs_pre + the evaluation) instead of four - and every iterate, status and count is the one of the four-launch rounds."""
    cands = [sc.make_candidate(70 + b, [64, 40, 64, 12, 64, 64][b], [16, 10, 16, 3, 16, 16][b], obstacles=(b % 2 == 0)) for b in range(6)]
    prob = frx.Problem(cands, sc.ZHANGJIAJIE, qd_intervals=16)
    prob.set_resident(False)
    x0 = prob.initial_guess()
    prob.set_eval_solo(0)
    ra = prob.optimize(1e-6, x0=x0, max_iterations=400)
    prob.set_eval_solo(2)
    rb = prob.optimize(1e-6, x0=x0, max_iterations=400)
    assert ra["resident"] == 0 and rb["resident"] == 0
    assert np.array_equal(ra["x"], rb["x"]) and np.array_equal(ra["objective"], rb["objective"])
    assert np.array_equal(ra["status"], rb["status"]) and np.array_equal(ra["iters"], rb["iters"]) and np.array_equal(ra["evals"], rb["evals"])
    assert ra["rounds"] == rb["rounds"]
    assert np.array_equal(ra["C"], rb["C"]) and np.array_equal(ra["T"], rb["T"])
    prob.close()


@pytest.mark.gpu
@pytest.mark.parametrize("N,gates,kappa", [(100, 25, 8), (12, 3, 70)])
def test_solo_does_not_apply_beyond_64_pieces_or_samples_per_piece(frx, sc, N, gates, kappa):
    """More than 64 pieces (the knot bodies' other geometry classes) or more than 64 quadrature samples per piece (a lane of the stage kernel then walks over several
    samples - found by running the whole GPU suite with FRX_EVAL_SOLO=1: kappa = 70 took the form and integrated 64 of its 71 samples): the handle keeps the stage kernels."""
    prob = frx.Problem([sc.make_candidate(80, N, gates)], sc.ZHANGJIAJIE, qd_intervals=kappa)
    assert prob.eval_solo() == 0 and not prob.solo_applies()
    with pytest.raises(frx.FrxError):
        prob.set_eval_solo(2)
    prob.close()
