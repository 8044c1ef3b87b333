"""HIP-graph replay of a detector's device program, keyed by static shapes (shared by the MQ-GLIP and MQ-GroundingDINO classes)."""
from collections import OrderedDict

import torch


class GraphRunner:
    """Mixin: expects `self._graphs` (OrderedDict), `self.graph_cache_size`, `self.graph_warm_calls`, `self.cache_stats`."""

    def _drop_graphs(self):
        """Release every captured graph together with its static buffers (the shared pool empties with them)."""
        for ent in list(getattr(self, "_graphs", {}).values()):
            ent.clear()
        self._graphs = OrderedDict()


    @staticmethod
    def _tree_map(fn, obj):
        if torch.is_tensor(obj):
            return fn(obj)
        if isinstance(obj, dict):
            return {k: GraphRunner._tree_map(fn, v) for k, v in obj.items()}
        if isinstance(obj, (list, tuple)):
            return type(obj)(GraphRunner._tree_map(fn, v) for v in obj)
        return obj

    @staticmethod
    def _tree_copy_(dst, src, seen=None, path=()):
        """Copy the caller's tensors into the graph's static input buffers.  `seen` (one dict per captured graph): a SMALL input that is the very
        tensor object of the previous replay, unmodified since (torch's version counter) -- the memoised host-side operands of a caption: token ids,
        masks, the query bank selection, the label -> token index, image sizes -- is already there: no copy launch.  Large inputs (the pixels, cached
        feature maps) are always copied."""
        if torch.is_tensor(dst):
            if seen is not None and src.numel() <= GraphRunner.SKIP_COPY_MAX_NUMEL:
                prev = seen.get(path)
                if prev is not None and prev[0] is src and prev[1] == src._version:
                    return
                seen[path] = (src, src._version)          # (a strong reference: an id cannot be recycled while it is the comparison's left side)
            dst.copy_(src, non_blocking=True)
        elif isinstance(dst, dict):
            for k in dst:
                GraphRunner._tree_copy_(dst[k], src[k], seen, path + (k,))
        elif isinstance(dst, (list, tuple)):
            for i, (d, s_) in enumerate(zip(dst, src)):
                GraphRunner._tree_copy_(d, s_, seen, path + (i,))

    SKIP_COPY_MAX_NUMEL = 1 << 20

    def _shape_key(self, obj):
        if torch.is_tensor(obj):
            return tuple(obj.shape)
        if isinstance(obj, dict):
            return tuple((k, self._shape_key(v)) for k, v in obj.items())
        if isinstance(obj, (list, tuple)):
            return tuple(self._shape_key(v) for v in obj)
        return obj

    def _run(self, program, inputs, use_graph):
        """Run `program(*inputs)`, as a HIP-graph replay when its static-shape key is warm.

        Key = the program and the SHAPES of its tensor inputs (+ the caption length in 16-token blocks, `inputs[-1]`: it selects the
        compile-time variant of the VLFuse image-side kernel, which only visits that many key blocks);
        image sizes are a tensor input (`im_wh`) and therefore not part of the key.  A key runs eagerly for its first
        `HIP_GRAPH_WARM_CALLS` calls (library autotuning, caches; cold keys never pay a capture), is captured on the next
        one and replayed afterwards.  At most `HIP_GRAPH_CACHE` graphs are kept (LRU): evicting one releases the graph,
        its static input / output buffers and its private activation pool.  The eager forward issues ~1500 launches
        (host-bound by ~17 ms / step at B = 8, profiles/r01_call3); a replay costs one launch."""
        fn = getattr(self, program)
        if not use_graph:
            self.cache_stats["eager"] += 1
            return fn(*inputs)
        key = (program, self._shape_key(inputs[:-1]), -(-int(inputs[-1]) // 16))
        ent = self._graphs.get(key)
        if ent is None:
            ent = self._graphs[key] = {"stage": 0, "calls": 0}
            while len(self._graphs) > max(1, self.graph_cache_size):
                _, old = self._graphs.popitem(last=False)
                old.clear()
                self.cache_stats["graph_evict"] += 1
        self._graphs.move_to_end(key)
        if ent["stage"] == 0:
            ent["calls"] += 1
            if ent["calls"] <= self.graph_warm_calls:
                self.cache_stats["eager"] += 1
                return fn(*inputs)
            static_in = self._tree_map(lambda t: t.clone(), inputs)
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            try:
                # thread_local: with torch.distributed initialised, the RCCL watchdog thread polls events concurrently;
                # under the default "global" mode that would invalidate the capture
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    static_out = fn(*static_in)
            except Exception as e:                              # keep running eagerly, but say so loudly
                import warnings
                warnings.warn(f"mq_det_amd: HIP graph capture failed ({type(e).__name__}: {e}); staying eager")
                ent.update(stage=-1)
                torch.cuda.synchronize()
                self.cache_stats["eager"] += 1
                return fn(*inputs)
            ent.update(stage=2, graph=g, inp=static_in, out=static_out, seen={})
            self.cache_stats["graph_capture"] += 1
        if ent["stage"] == -1:
            self.cache_stats["eager"] += 1
            return fn(*inputs)
        self._tree_copy_(ent["inp"], inputs, ent["seen"])
        ent["graph"].replay()
        self.cache_stats["graph_replay"] += 1
        return ent["out"]
