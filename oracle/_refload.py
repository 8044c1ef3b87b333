"""Import the reference's own Python modules in the build container (test infrastructure; used only
by oracle/gen_golden.py, never on the GPU box -- /root/reference does not exist there).

The reference (YifanXu74/MQ-Det) cannot be imported as a package here: it needs its compiled
`maskrcnn_benchmark._C` CUDA extension, timm, yacs, einops_exts, torchvision, pycocotools and a
transformers-4.x API.  What we do instead:
  * register *empty* package shells (`__path__` pointing at the real directories, `__init__` not
    executed) so that individual reference source files import by their real dotted names,
  * stub the missing third-party names that are identities at eval (timm DropPath, einops_exts
    rearrange_many, transformers doc decorators / moved helpers, yacs CfgNode),
  * stub the reference modules that only matter for training or data (rpn.loss, engine.inference,
    backbone.fbnet, clip_model),
  * `_C` is a module whose every attribute raises -- the two CUDA-only ops (DCNv2, ml_nms) are
    patched by the caller with the oracle's restatement, which is why those two stay "unpinned".
No reference source is copied; the files are executed from where they lie.
"""
import importlib
import logging
import math
import sys
import types

import torch.nn as nn

REF = "/root/reference"


class CfgNode(dict):
    """Just enough of yacs.config.CfgNode for config/defaults.py + YAML merging."""

    def __init__(self, init=None, new_allowed=False):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def merge(self, other):
        for k, v in other.items():
            if isinstance(v, dict) and isinstance(self.get(k), CfgNode):
                self[k].merge(v)
            else:
                if isinstance(v, str) and v[:1] in "([":      # yacs literal_eval's tuple / list strings
                    import ast
                    try:
                        v = ast.literal_eval(v)
                    except (ValueError, SyntaxError):
                        pass
                self[k] = CfgNode(v) if isinstance(v, dict) else v

    def merge_from_file(self, path):
        import yaml
        with open(path) as f:
            self.merge(yaml.safe_load(f))

    def clone(self):
        import copy
        return copy.deepcopy(self)

    def freeze(self):
        pass


class _DropPath(nn.Module):
    def __init__(self, p=0.0):
        super().__init__()

    def forward(self, x):
        return x


def _shell(name, path=None):
    m = types.ModuleType(name)
    m.__path__ = [path] if path else []
    sys.modules[name] = m
    return m


def _module(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


_loaded = None


def load():
    """Returns a namespace with the reference modules used for golden generation."""
    global _loaded
    if _loaded is not None:
        return _loaded
    # transformers must be imported before the timm stub exists (its lazy loader probes timm)
    import transformers.activations  # noqa: F401
    import transformers.modeling_utils as mu
    import transformers.models.bert.modeling_bert as mb
    import transformers.pytorch_utils as pu
    from einops import rearrange

    _shell("timm")
    _shell("timm.models")
    _module("timm.models.layers", DropPath=_DropPath,
            to_2tuple=lambda x: tuple(x) if isinstance(x, (tuple, list)) else (x, x),
            trunc_normal_=lambda t, std=1.0: nn.init.trunc_normal_(t, std=std))
    _module("einops_exts", rearrange_many=lambda ts, pat, **kw: [rearrange(t, pat, **kw) for t in ts])
    _shell("yacs")
    _module("yacs.config", CfgNode=CfgNode)
    for n in ("add_start_docstrings_to_model_forward", "add_code_sample_docstrings"):
        if not hasattr(mb, n):
            setattr(mb, n, lambda *a, **k: (lambda f: f))
    for n in ("BERT_INPUTS_DOCSTRING", "_CHECKPOINT_FOR_DOC", "_CONFIG_FOR_DOC"):
        if not hasattr(mb, n):
            setattr(mb, n, "{}")
    if not hasattr(mb, "logger"):
        mb.logger = logging.getLogger("ref")
    for n in ("apply_chunking_to_forward", "find_pruneable_heads_and_indices", "prune_linear_layer"):
        if not hasattr(mu, n):
            setattr(mu, n, getattr(pu, n, None))

    R = REF + "/maskrcnn_benchmark"
    for sub in ("", "modeling", "modeling/rpn", "modeling/backbone", "modeling/language_backbone",
                "modeling/query_selector", "layers", "utils", "structures", "engine", "config"):
        _shell(("maskrcnn_benchmark." + sub.replace("/", ".")).rstrip("."), R + ("/" + sub if sub else ""))

    def _no_c(*a, **k):
        raise RuntimeError("maskrcnn_benchmark._C is CUDA-only and not available")
    C = types.ModuleType("maskrcnn_benchmark._C")
    C.__getattr__ = lambda n: _no_c
    sys.modules["maskrcnn_benchmark._C"] = C
    sys.modules["maskrcnn_benchmark"]._C = C

    imp = importlib.import_module
    ns = types.SimpleNamespace()
    ns.defaults = imp("maskrcnn_benchmark.config.defaults")
    ns.swint = imp("maskrcnn_benchmark.modeling.backbone.swint")
    ns.fpn = imp("maskrcnn_benchmark.modeling.backbone.fpn")
    imp("maskrcnn_benchmark.utils.torch_dropout")
    ns.bert_new = imp("maskrcnn_benchmark.modeling.language_backbone.modeling_bert_new")
    imp("maskrcnn_benchmark.modeling.utils")
    ns.fuse_helper = imp("maskrcnn_benchmark.utils.fuse_helper")
    ns.rpn_bert = imp("maskrcnn_benchmark.modeling.rpn.modeling_bert")
    L = sys.modules["maskrcnn_benchmark.layers"]
    misc = imp("maskrcnn_benchmark.layers.misc")
    ns.dyrelu = imp("maskrcnn_benchmark.layers.dyrelu")
    ns.deform_conv = imp("maskrcnn_benchmark.layers.deform_conv")
    bn = imp("maskrcnn_benchmark.layers.batch_norm")
    se = imp("maskrcnn_benchmark.layers.se")
    L.Scale, L.Conv2d, L.DYReLU, L.SELayer = misc.Scale, misc.Conv2d, ns.dyrelu.DYReLU, se.SELayer
    L.ModulatedDeformConv = ns.deform_conv.ModulatedDeformConv
    L.NaiveSyncBatchNorm2d, L.FrozenBatchNorm2d = bn.NaiveSyncBatchNorm2d, bn.FrozenBatchNorm2d
    L.nms = L.ml_nms = _no_c
    _module("maskrcnn_benchmark.modeling.backbone.fbnet", math=math, __all__=["math"])
    _module("maskrcnn_benchmark.engine.inference",
            create_positive_map_label_to_token_from_positive_map=lambda *a, **k: None)
    _module("maskrcnn_benchmark.modeling.rpn.loss", make_atss_loss_evaluator=lambda *a, **k: None)
    _module("maskrcnn_benchmark.modeling.language_backbone.clip_model",
            QuickGELU=nn.GELU, LayerNorm=nn.LayerNorm, DropPath=_DropPath)
    imp("maskrcnn_benchmark.modeling.box_coder")
    ns.bounding_box = imp("maskrcnn_benchmark.structures.bounding_box")
    ns.boxlist_ops = imp("maskrcnn_benchmark.structures.boxlist_ops")
    ns.image_list = imp("maskrcnn_benchmark.structures.image_list")
    ns.inference = imp("maskrcnn_benchmark.modeling.rpn.inference")
    ns.anchor_generator = imp("maskrcnn_benchmark.modeling.rpn.anchor_generator")
    ns.vldyhead = imp("maskrcnn_benchmark.modeling.rpn.vldyhead")
    ns.query_selector = imp("maskrcnn_benchmark.modeling.query_selector.query_selector")
    _loaded = ns
    return ns


_gd_loaded = None


def load_gdino():
    """The reference's MQ-GroundingDINO modules (groundingdino_new/models/GroundingDINO/*), executed in place.  torchvision,
    the visualiser, the tokenizer download helpers, the training loss / matcher and the compiled `groundingdino_new._C` are
    stubbed (the reference's own pure-torch `multi_scale_deformable_attn_pytorch` runs instead of the CUDA op on CPU,
    ms_deform_attn.py:340-347)."""
    global _gd_loaded
    if _gd_loaded is not None:
        return _gd_loaded
    ns = load()
    imp = importlib.import_module
    G = REF + "/groundingdino_new"
    _module("torchvision", __version__="0.15.2", _is_tracing=lambda: False)
    _shell("torchvision.ops")
    _module("torchvision.ops.boxes", box_area=lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]), nms=None)
    _shell("torchvision.models")
    _module("torchvision.models._utils", IntermediateLayerGetter=object)
    for sub in ("", "util", "models", "models/GroundingDINO", "models/GroundingDINO/backbone"):
        _shell(("groundingdino_new." + sub.replace("/", ".")).rstrip("."), G + ("/" + sub if sub else ""))
    g = types.SimpleNamespace(base=ns)
    g.misc = imp("groundingdino_new.util.misc")
    g.utils = imp("groundingdino_new.models.GroundingDINO.utils")
    g.msda = imp("groundingdino_new.models.GroundingDINO.ms_deform_attn")
    g.fuse = imp("groundingdino_new.models.GroundingDINO.fuse_modules")
    g.vanilla = imp("groundingdino_new.models.GroundingDINO.transformer_vanilla")
    g.transformer = imp("groundingdino_new.models.GroundingDINO.transformer")
    g.bertwarper = imp("groundingdino_new.models.GroundingDINO.bertwarper")
    g.position_encoding = imp("groundingdino_new.models.GroundingDINO.backbone.position_encoding")
    g.swin = imp("groundingdino_new.models.GroundingDINO.backbone.swin_transformer")
    g.backbone = imp("groundingdino_new.models.GroundingDINO.backbone.backbone")
    sys.modules["groundingdino_new.models.GroundingDINO.backbone"].build_backbone = g.backbone.build_backbone
    imp("groundingdino_new.models.registry")
    _module("groundingdino_new.util.visualizer", COCOVisualizer=object)
    _module("groundingdino_new.util.utils", get_phrases_from_posmap=None)
    _module("groundingdino_new.util.vl_utils", create_positive_map_from_span=None)
    g.get_tokenlizer = _module("groundingdino_new.util.get_tokenlizer", get_tokenlizer=None, get_pretrained_language_model=None)
    _module("groundingdino_new.models.GroundingDINO.loss", SetCriterion=lambda **k: None)
    _module("groundingdino_new.models.GroundingDINO.matcher", build_matcher=lambda *a, **k: None)
    if "maskrcnn_benchmark.modeling.poolers" not in sys.modules:
        _module("maskrcnn_benchmark.modeling.poolers", CustomPooler=lambda **k: None, Pooler=lambda **k: None)
    sys.modules["maskrcnn_benchmark.modeling.language_backbone"].build_language_backbone = None
    sys.modules["maskrcnn_benchmark.modeling.query_selector"].build_query_selector = lambda cfg: ns.query_selector.QuerySelector(cfg)
    g.groundingdino = imp("groundingdino_new.models.GroundingDINO.groundingdino")
    _gd_loaded = g
    return g


def reference_cfg(*yaml_files):
    """The reference's own default config tree (config/defaults.py) + its YAMLs."""
    ns = load()
    cfg = ns.defaults._C.clone()
    for f in yaml_files:
        cfg.merge_from_file(REF + "/" + f)
    return cfg


def reference_functions(rel_path, names, extra_globals=None):
    """Execute selected top-level function definitions of a reference source file IN PLACE (nothing is copied), without
    importing the module around them -- e.g. the caption / positive-map builders of engine/inference.py, whose module
    imports the dataset stack.  Returns {name: function}."""
    import ast
    import os
    import re
    from collections import defaultdict

    import torch
    path = REF + "/" + rel_path
    src = open(path).read()
    tree = ast.parse(src)
    want = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    missing = set(names) - {n.name for n in want}
    if missing:
        raise KeyError(f"{rel_path}: no top-level function(s) {sorted(missing)}")
    mod = ast.Module(body=want, type_ignores=[])
    glb = {"re": re, "os": os, "torch": torch, "defaultdict": defaultdict}
    glb.update(extra_globals or {})
    exec(compile(mod, path, "exec"), glb)
    return {n: glb[n] for n in names}


def reference_classes(rel_path, class_names, function_names=(), extra_globals=None):
    """Like reference_functions, for top-level classes (plus helper functions they call) of a reference source file."""
    import ast
    import os
    import re
    from collections import defaultdict

    import torch
    path = REF + "/" + rel_path
    tree = ast.parse(open(path).read())
    want = [n for n in tree.body if (isinstance(n, ast.ClassDef) and n.name in class_names)
            or (isinstance(n, ast.FunctionDef) and n.name in function_names)]
    glb = {"re": re, "os": os, "torch": torch, "defaultdict": defaultdict}
    glb.update(extra_globals or {})
    exec(compile(ast.Module(body=want, type_ignores=[]), path, "exec"), glb)
    return {n: glb[n] for n in list(class_names) + list(function_names)}
