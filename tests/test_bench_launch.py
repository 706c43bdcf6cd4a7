"""bench.py --gpus N from a plain shell: the parent starts N rank processes itself."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parent_never_imports_torch_and_reports_failed_ranks(tmp_path):
    """Without a GPU every rank fails at its first device call; the parent must (a) have spawned exactly N
    ranks with the torchrun environment, (b) not have imported torch itself (it must never touch the GPU),
    (c) exit non-zero naming the failed ranks, with no JSON line on stdout."""
    probe = tmp_path / "sitecustomize.py"
    probe.write_text(
        "import os, sys\n"
        "if os.environ.get('RANK') is not None:\n"
        "    open(os.path.join(os.environ['PROBE_DIR'], 'rank%s' % os.environ['RANK']), 'w').write(\n"
        "        ' '.join(os.environ[k] for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')))\n"
        "    sys.exit(7)\n"   # stands for a rank that fails; no GPU is needed for the test
        "else:\n"
        "    import atexit\n"
        "    atexit.register(lambda: open(os.path.join(os.environ['PROBE_DIR'], 'parent'), 'w').write(\n"
        "        str('torch' in sys.modules)))\n")
    env = dict(os.environ, PYTHONPATH=str(tmp_path), PROBE_DIR=str(tmp_path))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "2"],
                         env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "ranks failed" in out.stderr
    assert out.stdout.strip() == ""
    assert (tmp_path / "parent").read_text() == "False"
    seen = sorted(f for f in os.listdir(tmp_path) if f.startswith("rank"))
    assert seen == ["rank0", "rank1", "rank2"]
    ports = set()
    for r, f in enumerate(seen):
        rank, local, world, addr, port = (tmp_path / f).read_text().split()
        assert (int(rank), int(local), int(world), addr) == (r, r, 3, "127.0.0.1")
        ports.add(port)
    assert len(ports) == 1
