"""the kernel SOURCES with fp32 operands, end to end, against the fp32 oracle at the north-star tolerance 1e-3.

On the device the product differs from the fp32 oracle by 2e-3 (backbone) ... 3e-2 (alignment logits) of the tensors' range -- DESIGN.md
section 7 attributes all of it to 16-bit operand rounding amplified by a randomly initialised network (the "operand floor").  This test
removes the rounding and nothing else: tests/simt compiles every kernel source a third time with `half_t = float` (entry points *_f32;
the emulated MFMA multiplies the floats exactly), the product's own host code (pipeline.py, ops.py wrappers, weight folding, plans) runs
unchanged on float32 tensors, and every stage of the tiny-depth full model, of the Swin / FPN stack and of one fusion layer at the FULL
800 x 1333 geometry must agree with the oracle to 1e-3.  Whatever error the device shows above that is therefore rounding, not logic.

TEST INFRASTRUCTURE ONLY (emulation library; the product raises without a GPU)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
CPU = torch.device("cpu")
_CXX = os.environ.get("SIMT_CXX", "/opt/rocm/lib/llvm/bin/clang++")
pytestmark = pytest.mark.skipif(not os.path.exists(_CXX), reason=f"{_CXX} not found: the kernel-source emulation cannot be built here")


# Round 6: the hot kernels of the split-precise build have ONE shape in both modes (planar tiles, C = 384 Swin MLP through the main kernel), so the second
# mode only repeats launches on the grouped offset conv and a few variants: it runs on request (MQ_SIMT_F32_MODES=1,2), not in the default CPU suite.
_MODES = [int(m) for m in os.environ.get("MQ_SIMT_F32_MODES", "1").split(",") if m]


@pytest.fixture(scope="module", params=_MODES, ids=[{1: "device_limits", 2: "every_kernel"}[m] for m in _MODES])
def f32(request):
    """params: 1 = the precise mode exactly as the MI355X runs it (160 KB of LDS per workgroup: ops.py picks the kernel variants that fit at
    twice the element size -- what passes here is what `MODEL.COMPUTE_DTYPE = "float32"` launches on the device); 2 = a 320 KB limit, i.e.
    every kernel in the shape the 16-bit modes launch it (the kernel LOGIC of the shipped fp16 / bf16 path against the fp32 oracle)."""
    import simt
    import parity_checks as pc
    from mq_det_amd.modeling import detector, pipeline

    def prepare(self, device=None):
        from mq_det_amd import ops
        ops.configure(self.cfg)
        self._kernels = dict(ops.KERNELS)
        self._validate_config()
        assert detector.compute_dtype(self.cfg) == torch.float32          # parity_checks builds its models with MODEL.COMPUTE_DTYPE = "float32"
        self._plan = pipeline.build_plan(self.state_dict(), self.cfg, CPU, dtype=torch.float32)
        self._plan_key, self.use_hip_graph = CPU, False
        return self._plan
    from mq_det_amd.modeling import gdino, gdino_pipeline as gp
    import gdino_checks as gc

    def prepare_gdino(self, device=None):
        from mq_det_amd import ops
        ops.configure(self.cfg)
        self._kernels = dict(ops.KERNELS)
        assert detector.compute_dtype(self.cfg) == torch.float32
        self._plan = gp.build_gdino_plan(self.state_dict(), self.cfg, CPU, self._swin, dtype=torch.float32)
        self._plan_key, self.use_hip_graph = CPU, False
        return self._plan
    saved = (detector.GeneralizedVLRCNN_New.prepare, pc.QUICK, pc.PINS, dict(pc._CACHE))
    saved_gd = (gdino.GroundingDINO.prepare, dict(gc._CACHE))
    detector.GeneralizedVLRCNN_New.prepare = prepare
    gdino.GroundingDINO.prepare = prepare_gdino
    gc._CACHE.clear()
    pc.QUICK, pc.PINS = True, False
    pc.use_dtype(torch.float32)
    pc._CACHE.clear()
    pc._f32_mode = request.param
    with simt.installed(f32=request.param):
        yield pc
    pc.use_dtype(torch.float16)
    detector.GeneralizedVLRCNN_New.prepare, pc.QUICK, pc.PINS = saved[:3]
    gdino.GroundingDINO.prepare = saved_gd[0]
    gc._CACHE.clear()
    gc._CACHE.update(saved_gd[1])
    pc._CACHE.clear()
    pc._CACHE.update(saved[3])


def _assert_ok(results, tol=1e-3):
    results = results if isinstance(results, list) else [results]
    assert results, "no results"
    sets = ("detections", "two-stage top-", "HIP-graph replay")            # set-valued rows carry a matched fraction, not a normalised error
    bad = [f"{r['name']}: norm_err {r['norm_err']:.2e} (tol {r['tol']:.1e})" for r in results if not r["ok"] or (r["tol"] > tol and not any(t in r["name"] for t in sets))]
    assert not bad, "\n".join(bad)
    return max(r["norm_err"] for r in results if not any(t in r["name"] for t in sets))


def test_tiny_full_model_meets_1e_3_with_fp32_operands(f32):
    """Swin -> FPN -> BERT + GCP -> 2 fusion layers (VLFuse, BERT, DyConv / DCNv2) -> heads -> class scores: every stage of
    check_full_model (the smoke() check) at 1e-3 of the reference's range, detections matched."""
    worst = _assert_ok(f32.check_full_model(CPU))
    print(f"worst normalised stage error with fp32 operands: {worst:.2e}")


@pytest.mark.parametrize("name", ["check_swin_fpn", "check_window_attention", "check_gcp_block", "check_pre_select", "check_vl_fuse", "check_dyconv",
                                  "check_align_fused", "check_bert_attn_qkv", "check_attention_text", "check_gcp_attn_fused", "check_dcn",
                                  "check_vlfuse_kernels", "check_swin_mlp"])
def test_blocks_meet_1e_3_with_fp32_operands(f32, name):
    # "every_kernel" only where the 16-bit modes launch something else than the device's precise mode does (the fused Swin MLP main kernel at
    # C = 384, the grouped offset conv, the double-buffered DCNv2, the VLFuse variants): the other kernels are the same launch in both modes
    if f32._f32_mode == 2 and name not in ("check_swin_fpn", "check_dyconv", "check_vl_fuse"):
        pytest.skip("same launch as under device_limits")
    _assert_ok(getattr(f32, name)(CPU))


def test_bert_layers_meet_1e_3_with_fp32_operands(f32):
    if f32._f32_mode == 2:
        pytest.skip("same launches as under device_limits")
    _assert_ok([f32.check_bert_layer(CPU, False), f32.check_bert_layer(CPU, True)])


def test_mq_glip_l_blocks_meet_1e_3_with_fp32_operands(f32):
    """Swin-L shapes (window 12, widths 192 / 384 / 768 / 1536) through the split-precise kernels: window attention and Swin + FPN at 1e-3."""
    if f32._f32_mode == 2:
        pytest.skip("same launches as under device_limits")
    _assert_ok(f32.check_window_attention(CPU, large=True))
    _assert_ok(f32.check_swin_fpn(CPU, large=True))


def test_groundingdino_meets_1e_3_with_fp32_operands(f32):
    """Round 6: MQ-GroundingDINO no longer refuses MODEL.COMPUTE_DTYPE = float32 -- the MSDeformAttn kernels have their *_f32
    twin (the fused-query form's `qproj` is a float there) and every other kernel of the family already had one.  The masked attention / fusion kernels of the
    family and the shallow whole model with vision queries against the oracle: every stage at 1e-3 of its range."""
    if f32._f32_mode == 2:
        pytest.skip("same launches as under device_limits")
    import gdino_checks as gc
    _assert_ok(gc.check_attention_qk_mask(CPU))
    _assert_ok(gc.check_vlfuse_heads_mask(CPU))
    # (the shallow model with vision queries runs every kernel of the family incl. both MSDeformAttn forms; the stand-alone sampling check at the
    # 22 323-query encoder shape and the text-only B = 2 model run in the GPU suite -- through the emulation they alone took four minutes)
    worst = _assert_ok(gc.check_gdino_model(CPU, vq=True, graph=False))
    print(f"MQ-GroundingDINO (shallow) with fp32 operands: worst normalised stage error {worst:.2e}")


def test_one_fusion_layer_at_the_benchmark_geometry_meets_1e_3_with_fp32_operands(f32):
    """VLFuse + clamped BERT layer + DyConv on the pyramid tokens of one image, 141 live text tokens: 1e-3 at every output.  Default: the
    pyramid of a 400 x 672 image (5 577 tokens, every level boundary, ragged last tiles; ~20 s); MQ_SIMT_FULL=1: the 22 400 tokens of an
    800 x 1333 image, every tile / level boundary of the benchmark shape (83 s; measured worst error 8.5e-6)."""
    import time
    t0 = time.time()
    full = os.environ.get("MQ_SIMT_FULL", "0") == "1"
    sizes = ((100, 168), (50, 84), (25, 42), (13, 21), (7, 11)) if full else ((50, 84), (25, 42), (13, 21), (7, 11), (4, 6))
    worst = _assert_ok(f32.check_fusion_layer(CPU, sizes=sizes))
    print(f"fusion layer at {'full' if full else 'half'} geometry: worst normalised error {worst:.2e} ({time.time() - t0:.0f} s through the emulation)")
