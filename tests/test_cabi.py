"""CPU: the C-ABI shared library builds, loads and exports every symbol include/epipolar_hip.h declares
(no compute calls without a GPU), and the product package has no route into the oracle."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from epipolarpose_amd import build, hip
    build.build(verbose=False)
    return hip.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "epipolar_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(epi_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(lib):
    from epipolarpose_amd import hip
    names = declared_symbols()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), "symbol %s declared in the header but not exported" % n
    assert set(names) == set(hip.EXPORTED_SYMBOLS), "ctypes signature table and header disagree"


def test_host_only_entry_points(lib):
    assert b"gfx950" in lib.epi_version()
    assert lib.epi_status_string(0) == b"ok"
    assert lib.epi_status_string(3) == b"workspace too small"
    # 544 rows x 1 chunk (scalar-fallback sizing: 262144 / 4096 = 64 chunks) x 32 B
    assert lib.epi_softargmax3d_workspace_bytes(32, 17, 64, 64, 64) == 32 * 17 * 64 * 32
    assert lib.epi_softargmax3d_workspace_bytes(0, 17, 64, 64, 64) == 0
    assert lib.epi_argmax_workspace_bytes(4, 100) == 4 * 16
    from epipolarpose_amd import hip
    assert all(lib.epi_bn_sum_copies(c) == hip.bn_sum_copies(c) for c in (8, 64, 256, 512, 1024, 2048))
    # argument validation happens before any device work
    assert lib.epi_softargmax3d_fwd(None, 0, 0, 1, 1, 1, 1, 1, None, None, None, None, 0, None) == 1
    assert lib.epi_joint_loss(None, None, None, 1, 3, 0, 0, 1, None, None, None) == 1
    assert lib.epi_triangulate_dlt(None, 2, None, 2, 1, 2, 1, None, None, None) == 1


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "epipolarpose_amd")
    offenders = []
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b|from\s+\.+\s*import\s+oracle|oracle/", src, flags=re.M):
                    offenders.append(os.path.join(dirpath, f))
    assert not offenders, offenders
    for f in ("bench.py",):
        p = os.path.join(ROOT, f)
        if os.path.exists(p):
            src = open(p).read()
            for m in re.finditer(r"^.*\boracle\b.*$", src, flags=re.M):
                line = m.group(0)
                assert "cpu_baseline" in src[max(0, m.start() - 3000):m.end()], "oracle used outside cpu_baseline: " + line


def test_no_cpu_fallback():
    import torch
    from epipolarpose_amd import hip
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        hip.softargmax3d_fwd(torch.zeros(1, 8, 2, 2), 2)


def test_torch_glue_builds_loads_and_links_the_c_abi():
    """csrc/torch_glue.cpp (C++ autograd glue) compiles against the installed torch, loads without a GPU, reports the
    library's version and refuses CPU tensors (no CPU fallback)."""
    import pytest
    import torch
    from epipolarpose_amd import build, hip
    build.build(verbose=False)
    build.build_glue(verbose=False)
    g = hip.glue()
    assert g.abi_version() == hip.load().epi_version().decode()
    from epipolarpose_amd.models.fused import FusedBatchNormAct
    m = FusedBatchNormAct(8)
    with pytest.raises(RuntimeError, match="GPU"):
        m(torch.zeros(2, 8, 4, 4, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last))


def declared_prototypes():
    """{name: number of parameters} for every function declared in include/epipolar_hip.h."""
    text = open(os.path.join(ROOT, "include", "epipolar_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(epi_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
    return out


def test_ctypes_signatures_match_the_header_arity():
    """A wrong argument count in the ctypes table corrupts the call silently (e.g. a stream pointer read as garbage):
    the number of parameters of every prototype in the header must equal the length of its ctypes argtypes."""
    from epipolarpose_amd import hip
    protos = declared_prototypes()
    assert set(protos) == set(hip._SIGNATURES)
    for name, (res, args) in hip._SIGNATURES.items():
        assert protos[name] == len(args), "%s: header declares %d parameters, ctypes table has %d" % (name, protos[name], len(args))


def test_glue_calls_use_the_header_prototypes():
    """csrc/torch_glue.cpp includes the public header (the compiler checks its calls); this guards against a private re-declaration."""
    src = open(os.path.join(ROOT, "epipolarpose_amd", "csrc", "torch_glue.cpp")).read()
    assert '#include "../../include/epipolar_hip.h"' in src
    assert not re.search(r'extern\s+"C"\s+int\s+epi_', src)


def test_weight_gradient_plan_is_consistent_with_its_workspace():
    """epi_gemm_tn_plan (host only): the slab workspace the library asks for is exactly splits x result x 4 bytes, a split covers whole
    64-row reduction tiles and all splits together cover R; the deep layers of ResNet-50 at batch 32 as pinned examples."""
    from epipolarpose_amd import hip
    lib = hip.load()
    for r, i, j, ntap in ((131072, 64, 64, 1), (131072, 64, 64, 9), (32768, 512, 128, 1), (8192, 256, 256, 9), (2048, 2048, 512, 1), (2048, 512, 512, 9),
                          (131072, 1088, 256, 1), (8192, 2048, 256, 16), (512, 64, 64, 1)):
        p = hip.gemm_tn_plan(r, i, j, ntap)
        assert p["cfg"] in (0, 1, 2) and p["tiles"] >= 1 and p["nsplit"] >= 1
        assert lib.epi_gemm_tn_workspace_bytes(r, i, j, ntap) == p["slab_bytes"]
        assert p["rows_per_split"] % 64 == 0 and p["nsplit"] * p["rows_per_split"] >= r > (p["nsplit"] - 1) * p["rows_per_split"]
        if i <= 64:
            assert p["cfg"] == 1                     # the narrow tile serves the 64-channel layers
    assert hip.gemm_tn_plan(2048, 512, 512, 9) == {"cfg": 0, "tiles": 144, "nsplit": 2, "rows_per_split": 1024, "slab_bytes": 2 * 512 * 4608 * 4}
    # round 4: long reductions over large operands (the final layer, the last deconvolution) take the 256 x 256 tile; shorter ones do not
    assert hip.gemm_tn_plan(131072, 1088, 256, 1)["cfg"] == 2 and hip.gemm_tn_plan(32768, 256, 256, 16)["cfg"] == 2
    assert hip.gemm_tn_plan(8192, 256, 256, 16)["cfg"] == 0 and hip.gemm_tn_plan(2048, 2048, 256, 16)["cfg"] == 0


def test_grouped_weight_gradient_plan():
    """epi_wgrad_group_plan (host only): a ResNet-50 stage's weight gradients together need no (or hardly any) reduction split -- the deep
    stages at batch 32 stay unsplit, the whole backbone parks < 100 MB of fp32 slabs per step (866 MB with one launch per layer) -- and
    the slab bytes are exactly the 256-byte-aligned sum of splits x result x 4 over the split items."""
    from epipolarpose_amd import hip
    assert hip.wgrad_group_max() >= 24
    b = 32
    layer4 = [("conv", b, 8, 8, 2048, 512, 1, 1, 0), ("conv", b, 8, 8, 512, 512, 3, 1, 1), ("conv", b, 8, 8, 512, 2048, 1, 1, 0)] * 2 + \
             [("conv", b, 16, 16, 1024, 512, 1, 1, 0), ("conv", b, 16, 16, 512, 512, 3, 2, 1), ("conv", b, 8, 8, 512, 2048, 1, 1, 0), ("conv", b, 16, 16, 1024, 2048, 1, 2, 0)]
    slab, ns = hip.wgrad_group_plan(layer4)
    assert ns[:6] == [1] * 6 and ns[7:] == [1] * 3 and slab <= 8 << 20, (slab, ns)
    layer3 = [("conv", b, 16, 16, 1024, 256, 1, 1, 0), ("conv", b, 16, 16, 256, 256, 3, 1, 1), ("conv", b, 16, 16, 256, 1024, 1, 1, 0)] * 5 + \
             [("conv", b, 32, 32, 512, 256, 1, 1, 0), ("conv", b, 32, 32, 256, 256, 3, 2, 1), ("conv", b, 16, 16, 256, 1024, 1, 1, 0), ("conv", b, 32, 32, 512, 1024, 1, 2, 0)]
    slab3, ns3 = hip.wgrad_group_plan(layer3)
    assert ns3[:15] == [1] * 15 and slab3 <= 4 << 20, (slab3, ns3)
    # alone, one of these layers is cut into 8 .. 16 slices
    assert hip.gemm_tn_plan(8192, 256, 1024, 1)["nsplit"] >= 8
    one, ns1 = hip.wgrad_group_plan([("conv", b, 64, 64, 64, 64, 3, 1, 1)])
    assert ns1[0] > 1 and one == (ns1[0] * 64 * 576 * 4 + 255) // 256 * 256
    lib = hip.load()
    assert lib.epi_wgrad_group_plan(None, 0, None, None) == 1          # argument validation before any work


def test_grouped_weight_gradient_launches_fit_the_chip_in_one_round():
    """Round 4: the split of a grouped launch is chosen by the makespan of its workgroups on the chip's resident slots (512 workgroups of the
    128 x 128 class, 768 of the 64 x 128 class on 256 CUs).  For the two stages of ResNet-50 at batch 32 whose items are ALL split (layers 1 and 2:
    every workgroup has the same length) a launch of slightly more workgroups than slots runs a second round that doubles its time -- the
    round-3 model chose 640 and 860 workgroups there (measured 271 / 144 us; 200 / 110 us with 476 / 740)."""
    from epipolarpose_amd import hip
    b = 32

    def workgroups(shapes, ns):
        wg = [0, 0]
        for (_, _, _, _, cin, cout, k, _, _), v in zip(shapes, ns):
            cls = 1 if cout <= 64 else 0
            wg[cls] += -(-cout // (64 if cls else 128)) * -(-(k * k * cin) // 128) * v
        return wg
    layer1 = [("conv", b, 64, 64, 256, 64, 1, 1, 0), ("conv", b, 64, 64, 64, 64, 3, 1, 1), ("conv", b, 64, 64, 64, 256, 1, 1, 0)] * 2 + \
             [("conv", b, 64, 64, 64, 64, 1, 1, 0), ("conv", b, 64, 64, 64, 64, 3, 1, 1), ("conv", b, 64, 64, 64, 256, 1, 1, 0), ("conv", b, 64, 64, 64, 256, 1, 1, 0)]
    _, ns = hip.wgrad_group_plan(layer1)
    wide, narrow = workgroups(layer1, ns)
    assert 384 <= wide <= 512 and 576 <= narrow <= 768, (wide, narrow, ns)
    layer2 = [("conv", b, 32, 32, 512, 128, 1, 1, 0), ("conv", b, 32, 32, 128, 128, 3, 1, 1), ("conv", b, 32, 32, 128, 512, 1, 1, 0)] * 3 + \
             [("conv", b, 64, 64, 256, 128, 1, 1, 0), ("conv", b, 64, 64, 128, 128, 3, 2, 1), ("conv", b, 32, 32, 128, 512, 1, 1, 0), ("conv", b, 64, 64, 256, 512, 1, 2, 0)]
    _, ns2 = hip.wgrad_group_plan(layer2)
    wide2, narrow2 = workgroups(layer2, ns2)
    assert narrow2 == 0 and 384 <= wide2 <= 512, (wide2, ns2)
    # the same question asked twice gives the same plan (the choice is remembered per shape set)
    assert hip.wgrad_group_plan(layer2)[1] == ns2


def test_gemm_plan_prefers_unsplit_128_tiles_over_a_k_split_256_plan():
    """Round 4 (host only, through the workspace the library asks for): a GEMM whose 256 x 256 plan would have to split K while 128 x 128 tiles fill
    the chip unsplit needs NO split-K slabs any more -- the first deconvolution's backward-data of the head, 2048 x 2048 x 4096 -- while the
    shapes that have too few 128 x 128 tiles to fill the chip (layer 4's 2048 x 512 x 2048; the first deconvolution's forward, four phases of
    2048 x 256 x 8192) still split."""
    from epipolarpose_amd import hip
    lib = hip.load()
    assert lib.epi_gemm_workspace_bytes(2048, 2048, 4096, 1) == 0
    assert lib.epi_gemm_workspace_bytes(2048, 512, 2048, 1) == 4 * 2048 * 512 * 4
    assert lib.epi_gemm_workspace_bytes(2048, 256, 8192, 4) > 0
