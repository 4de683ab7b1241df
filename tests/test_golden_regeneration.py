"""The committed golden vectors ARE the reference's outputs: with /root/reference present (the build container), tests/golden/make_golden.py -- which
imports and EXECUTES the reference's own modules on seeded inputs -- is re-run for the fast sets into a temporary directory and every array must come out
bit-identical to the committed .npz files.  Skipped where the reference is absent (the GPU box): there the fixtures are all there is."""
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FAST_SETS = ["integral", "triangulation", "geometry", "maxpreds", "evaluation"]


@pytest.mark.skipif(not os.path.isdir("/root/reference/lib"), reason="the reference repository is not present on this host")
def test_fast_golden_sets_regenerate_bit_identically(tmp_path):
    env = dict(os.environ, EPI_GOLDEN_OUT=str(tmp_path))
    run = subprocess.run([sys.executable, os.path.join(HERE, "golden", "make_golden.py")] + FAST_SETS, env=env, capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    n_arrays = 0
    for name in FAST_SETS:
        new, old = np.load(str(tmp_path / (name + ".npz")), allow_pickle=False), np.load(os.path.join(HERE, "golden", name + ".npz"), allow_pickle=False)
        assert sorted(new.files) == sorted(old.files), name
        for k in old.files:
            a, b = new[k], old[k]
            assert a.dtype == b.dtype and a.shape == b.shape, (name, k)
            assert a.tobytes() == b.tobytes(), (name, k)
            n_arrays += 1
    assert n_arrays >= 200, n_arrays


@pytest.mark.skipif(not os.path.isdir("/root/reference/lib"), reason="the reference repository is not present on this host")
def test_voc_fixture_and_occluders_regenerate_bit_identically(tmp_path):
    """tests/golden/make_voc_fixture.py writes the Pascal-VOC tree and runs the reference's own ``load_occluders`` on it: the tree's files and the
    occluder arrays must come out byte-identical to the committed ones."""
    env = dict(os.environ, EPI_GOLDEN_OUT=str(tmp_path))
    run = subprocess.run([sys.executable, os.path.join(HERE, "golden", "make_voc_fixture.py")], env=env, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    new, old = np.load(str(tmp_path / "voc_occluders.npz")), np.load(os.path.join(HERE, "golden", "voc_occluders.npz"))
    assert sorted(new.files) == sorted(old.files)
    for k in old.files:
        assert new[k].dtype == old[k].dtype and new[k].tobytes() == old[k].tobytes(), k
    n_files = 0
    for d, _, files in os.walk(os.path.join(HERE, "golden", "voc_fixture")):
        for f in files:
            rel = os.path.relpath(os.path.join(d, f), os.path.join(HERE, "golden", "voc_fixture"))
            with open(os.path.join(d, f), "rb") as a, open(str(tmp_path / "voc_fixture" / rel), "rb") as b:
                assert a.read() == b.read(), rel
            n_files += 1
    assert n_files == 12, n_files                 # 4 annotations + 4 JPEG + 3 segmentations + the README
