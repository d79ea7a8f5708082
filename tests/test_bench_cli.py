"""bench.py's rank handling (no GPU needed): a run that cannot have the ranks --gpus asks for must fail loudly,
never print a line with fewer GPUs."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=300)


def test_more_gpus_than_the_node_has_is_refused():
    import torch

    n = torch.cuda.device_count() + 1
    r = _run(["--gpus", str(max(n, 2)), "--steps", "1", "--warmup", "0"])
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert "refusing" in r.stderr and not r.stdout.strip()


def test_launcher_world_size_must_equal_gpus():
    r = _run(["--gpus", "2", "--steps", "1"], {"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE=4" in r.stderr and not r.stdout.strip()
    r = _run(["--gpus", "8", "--steps", "1"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and not r.stdout.strip()


def test_gpu_leg_of_bench_does_not_import_the_oracle():
    """Only cpu_baseline() may touch oracle/ (it is the checker / baseline, never the product or its data)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    head, _, tail = src.partition("def cpu_baseline(")
    body, _, rest = tail.partition("\ndef ")
    assert "oracle" not in head.split('"""', 2)[2] and "from oracle" not in rest and "import oracle" not in rest
    assert "from oracle" in body


def test_default_workloads_strong_scaling_on_the_200_image_set():
    """--gpus N > 1 measures north_star's experiment (the fixed 200-image set, strong scaling); one GPU
    keeps configs[1] (50 images); --weak keeps the old 50-per-GPU experiment; c5 is one GPU's share of configs[4]."""
    sys.path.insert(0, ROOT)
    import bench

    assert bench.plan_job("c2", 1) == (50, "strong")
    for n in (2, 4, 8):
        assert bench.plan_job("c2", n) == (200, "strong")
        assert bench.plan_job("c2", n, weak=True) == (50 * n, "weak")
    assert bench.plan_job("c2", 1, images_total=200) == (200, "strong")      # the N = 1 anchor of the scaling curve
    assert bench.plan_job("c5", 1) == (63, "strong") and bench.plan_job("c5", 8) == (504, "weak")
    import pytest

    with pytest.raises(ValueError):
        bench.plan_job("c2", 2, images_total=100, weak=True)
    with pytest.raises(ValueError):
        bench.plan_job("c2", 8, images_total=10)


def test_contradicting_flags_are_refused_before_any_gpu_call():
    r = _run(["--gpus", "1", "--weak", "--images-total", "100"])
    assert r.returncode == 2 and "exclude" in r.stderr and not r.stdout.strip()
    r = _run(["--gpus", "1", "--images-total", "1"])
    assert r.returncode == 2 and not r.stdout.strip()
