"""Rulebooks of a whole chain of sparse layers from the input coordinates alone, in ONE call of the compiled binding
(binding.cpp geometry_walk) -- what SparseConvolution.forward_geometry does layer by layer in Python.  The plan (which layer
builds, which reuses an earlier layer's rulebook through its indice_key or through an identical geometry on the same level,
which inverse layer returns to which level) is derived once per (layer list, input shape); running it fills the
indice_dict / geometry cache exactly as the layer-by-layer walk would, so a later forward finds every rulebook ready."""
from . import ops
from .conv import SparseConvolution
from .modules import SparseSequential


def flatten_convs(*stages):
    """the SparseConvolution layers of (nested) SparseSequential containers, in execution order"""
    out = []
    for s in stages:
        if isinstance(s, SparseConvolution):
            out.append(s)
        elif isinstance(s, SparseSequential):
            out.extend(m for m in s._flat_modules() if isinstance(m, SparseConvolution))
        else:
            raise TypeError("geometry plan: %s is neither a sparse conv nor a SparseSequential" % type(s).__name__)
    return out


def _sig(conv):
    k, d = tuple(int(v) for v in conv.kernel_size), tuple(int(v) for v in conv.dilation)
    if conv.subm:
        return (k, d, True, False)
    return (k, d, False, bool(conv.transposed), tuple(int(v) for v in conv.stride), tuple(int(v) for v in conv.padding),
            tuple(int(v) for v in conv.output_padding))


class GeometryPlan(object):
    def __init__(self, convs, spatial_shape, batch_size):
        self.convs, self.batch_size = list(convs), int(batch_size)
        self.entries = []          # per layer: (kind, ref, geometry or None, in_shape, out_shape)
        by_key, by_sig, level, next_level = {}, {}, 0, 1
        levels = []                # per layer (level in, level out)
        shape = [int(v) for v in spatial_shape]
        for i, conv in enumerate(self.convs):
            if conv.conv1x1 or conv.ndim != 3:
                raise ValueError("geometry plan: only 3-D, non-1x1 sparse convs")
            if conv.inverse:
                r = by_key[conv.indice_key]
                self.entries.append((2, r, None, shape, self.entries[r][3]))
                levels.append((level, levels[r][0]))
                level, shape = levels[r][0], self.entries[r][3]
                continue
            r = by_key.get(conv.indice_key) if conv.indice_key is not None else None
            if r is None:
                r = by_sig.get((level, _sig(conv)))
            if r is not None:       # cached rulebook (used without checking the layer's own geometry)
                self.entries.append((3, r, None, shape, self.entries[r][4]))
                levels.append((level, levels[r][1]))
                level, shape = levels[r][1], self.entries[r][4]
            else:
                g = conv._geometry(shape)
                out_shape = [int(v) for v in conv._out_shape(shape)]
                out_level = level if conv.subm else next_level
                next_level += 0 if conv.subm else 1
                self.entries.append((0 if conv.subm else 1, -1, g, shape, out_shape))
                levels.append((level, out_level))
                by_sig[(level, _sig(conv))] = i
                level, shape = out_level, out_shape
            if conv.indice_key is not None and conv.indice_key not in by_key:
                by_key[conv.indice_key] = i if self.entries[i][0] < 2 else self.entries[i][1]
        e = self.entries
        self.args = ([x[0] for x in e], [x[2].a_in if x[2] else 0 for x in e], [x[2].a_out if x[2] else 0 for x in e],
                     [x[2].a_k if x[2] else 0 for x in e], [x[2].a_s if x[2] else 0 for x in e], [x[2].a_p if x[2] else 0 for x in e],
                     [x[2].a_d if x[2] else 0 for x in e], [x[2].mode if x[2] else 0 for x in e], [x[2].K if x[2] else 0 for x in e],
                     [ops._conv_ws_bytes(x[2], self.batch_size) if (x[2] is not None and not x[2].subm) else 0 for x in e],
                     [x[1] for x in e])

    def start(self, indices, side_stream=True):
        """phase A of the walk (the levels + the read-back of their row counts) forked onto the walk's side stream; finish() joins it.
        Whatever is launched in between overlaps with it without a host wait.  (side_stream=False: on the current stream with the
        blocking read-back at once -- run() in two calls.)"""
        return ops.fast().geometry_walk_start(indices, self.batch_size, *self.args, 1 if side_stream else 0)

    def finish(self, handle, indices, indice_dict, have=None):
        """join start(): size and fill the maps on the current stream and file the rulebooks as run() does.  have: {layer
        index: Rulebook} for built submanifold layers whose rulebook the caller made itself (their maps are not built again)."""
        have = have or {}
        built = ops.fast().geometry_walk_finish(handle, sorted(have))
        return self._file(built, indices, indice_dict, have)

    def run(self, indices, indice_dict):
        """build every rulebook of the plan for `indices` and file them in indice_dict (by indice_key) and in its geometry cache"""
        F = ops.fast()
        prof = ops.PROFILE
        if prof is not None:   # one span for the whole chain: two phases, one read-back (csrc/rulebook.hip)
            import torch
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        built = F.geometry_walk(indices, self.batch_size, *self.args)
        if prof is not None:
            e1.record()
            nbytes, rows, pairs, n_rb = 0, 0, 0, 0
            for b in built:
                if b:   # SURVEY.md section 8d, rulebook: 16 N_in + 16 N_out + 8 sum_k P_k bytes
                    p_k = ops._num_pairs(b[2])
                    nbytes += 16 * b[0].shape[0] + 16 * b[1].shape[0] + 8 * p_k
                    rows += b[1].shape[0]
                    pairs += p_k
                    n_rb += 1
            prof.records.append(("rulebook", e0, e1, nbytes, 0, dict(rows=rows, pairs=pairs, rulebooks=n_rb, chain=True)))
        return self._file(built, indices, indice_dict, {})

    def _file(self, built, indices, indice_dict, have):
        geom = indice_dict.setdefault("__geometry_cache__", {})
        rbs = [None] * len(self.convs)
        for i, (conv, (kind, ref, g, in_shape, out_shape)) in enumerate(zip(self.convs, self.entries)):
            if i in have:
                rb = rbs[i] = have[i]
                geom[conv._gkey(rb.in_indices, in_shape)] = (rb, rb.in_indices)
            elif kind < 2:
                in_idx, out_idx, nbr_out, nbr_in = built[i][:4]
                o_out, o_in = (built[i][4], built[i][5]) if (len(built[i]) > 4 and ops.ROW_ORDER) else (None, None)
                rb = rbs[i] = ops.Rulebook(out_idx, in_idx, nbr_out, nbr_in, g.in_list, g.out_list, g.K, g.mode, o_out, o_in)
                geom[conv._gkey(in_idx, in_shape)] = (rb, in_idx)
            else:
                rb = rbs[i] = rbs[ref]
            if kind != 2 and conv.indice_key is not None and conv.indice_key not in indice_dict:
                indice_dict[conv.indice_key] = rb
        return rbs
