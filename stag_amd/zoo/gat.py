"""GAT layer (reference: stag/zoo/gat.py:7-149; DGL GATConv layout).  The edge
weight has one column per head (`sample_dimension = num_heads`, :11) and scales the
leaky-relu'd logit BEFORE the edge softmax (:117-119).  Logits, noise, softmax and
the weighted sum are one fused kernel (`ops.gat_aggregate`)."""
import torch

from .. import ops


class GAT(torch.nn.Module):
    supports_edge_noise = True
    supports_edge_noise_grad = True   # vi=True: ops.gat_aggregate forms the [E, H] weights from the descriptor

    def __init__(self, in_feats, out_feats, num_heads=4, feat_drop=0.0, attn_drop=0.0,
                 negative_slope=0.2, residual=False, activation=None, allow_zero_in_degree=False,
                 bias=True, last=False):
        super().__init__()
        self._num_heads, self._in_feats, self._out_feats = num_heads, in_feats, out_feats
        self._negative_slope = negative_slope
        self.fc = torch.nn.Linear(in_feats, out_feats * num_heads, bias=False)
        self.attn_l = torch.nn.Parameter(torch.empty(1, num_heads, out_feats))
        self.attn_r = torch.nn.Parameter(torch.empty(1, num_heads, out_feats))
        self.feat_drop = torch.nn.Dropout(feat_drop)
        self.attn_drop = torch.nn.Dropout(attn_drop)     # stag/zoo/gat.py:122: dropout on the edge softmax
        self.bias = torch.nn.Parameter(torch.empty(num_heads * out_feats)) if bias else None
        if residual:
            self.res_fc = (torch.nn.Linear(in_feats, num_heads * out_feats, bias=False)
                           if in_feats != num_heads * out_feats else torch.nn.Identity())
        else:
            self.res_fc = None
        self.activation = activation
        self.last = last
        self.sample_dimension = num_heads
        # the generator the in-kernel attention dropout draws from: a StagLayer hands over its own before every
        # forward, so saving / restoring the layer's generator replays the masks too; None: the default generator
        self.noise_generator = None
        self.reset_parameters()

    def reset_parameters(self):
        gain = torch.nn.init.calculate_gain("relu")
        torch.nn.init.xavier_uniform_(self.fc.weight, gain=gain)
        torch.nn.init.xavier_uniform_(self.attn_l, gain=gain)
        torch.nn.init.xavier_uniform_(self.attn_r, gain=gain)
        if self.bias is not None:
            torch.nn.init.constant_(self.bias, 0)
        if isinstance(self.res_fc, torch.nn.Linear):
            torch.nn.init.xavier_uniform_(self.res_fc.weight, gain=gain)

    def extra_offsets_per_forward(self):
        """Generator offsets a forward takes besides the edge noise: one for the in-kernel attention-dropout mask."""
        H, F = self._num_heads, self._out_feats
        Fp = self._padded_width(H, F)
        return int(self.training and self.attn_drop.p > 0.0 and ops.attn_drop_fusable(H, Fp, ops.DEFAULT_SEG_LEN))

    @staticmethod
    def _padded_width(H, F):
        """F if the cooperative kernels take it (F % 4 == 0: a head gets F / 4 rounded up to a power of two lanes,
        the extra lanes idle) or no padded width fits them; else the next multiple of 4."""
        if F % 4 == 0:
            return F
        Fp = (F + 3) // 4 * 4
        return Fp if ops.gat_cooperative_shape(H, Fp, ops.DEFAULT_SEG_LEN) else F

    def forward(self, graph, feat, get_attention=False, edge_weight=None):
        H, F = self._num_heads, self._out_feats
        h = self.feat_drop(feat)
        # Head widths that are not a multiple of 4 — class counts like 7, 121 on the last layer of the reference's
        # GAT scripts — run with each head zero-padded to the next multiple of 4: the padding is rows of zeros in
        # the fc weight and zeros in attn_l / attn_r (parameter-sized ops), so the extra channels of ft are exactly
        # 0, add nothing to the logits, come out as 0 and are cut off again; the layer keeps the fused forward, the
        # one-gather backward and in-kernel attention dropout instead of the composed path.  (F % 4 == 0 with F / 4
        # not a power of two — 40 classes on arxiv, scripts/arxiv_mle/gat/run.py:50-58 — needs no padding: the
        # kernels give such a head the next power of two lanes and leave the extra ones idle.)
        Fp = self._padded_width(H, F) if feat.is_cuda else F
        if Fp != F:
            pad = torch.nn.functional.pad
            wp = pad(self.fc.weight.view(H, F, -1), (0, 0, 0, Fp - F)).reshape(H * Fp, -1)
            attn_l, attn_r = pad(self.attn_l, (0, Fp - F)), pad(self.attn_r, (0, Fp - F))
            F_out, F = F, Fp
        else:
            wp, attn_l, attn_r, F_out = self.fc.weight, self.attn_l, self.attn_r, F
        ft = ops.node_linear(h, wp.t()).view(-1, H, F)
        # el[n,h] = sum_f ft[n,h,f] attn_l[h,f], er likewise (zoo/gat.py:109-110).  The elementwise-multiply +
        # reduce form costs 2 x 242 us at cfg5 (and as much again in the backward); ops.head_dot is one pass
        # over ft forward and one back.  Head widths it does not take go through ONE [N, HF] x [HF, 2H]
        # product with a block-diagonal right side.
        lr = ops.head_dot(ft, attn_l, attn_r)
        if lr is not None:
            el, er = lr
        else:
            eye = torch.eye(H, dtype=ft.dtype, device=ft.device)
            w_lr = torch.cat([(attn_l.reshape(H, F, 1) * eye.reshape(H, 1, H)).reshape(H * F, H),
                              (attn_r.reshape(H, F, 1) * eye.reshape(H, 1, H)).reshape(H * F, H)], 1)
            elr = ops.node_linear(ft.reshape(-1, H * F), w_lr)
            el, er = elr[:, :H], elr[:, H:]
        if edge_weight is not None:
            assert edge_weight.shape[0] == graph.number_of_edges()
        # attention dropout (the reference's scripts train with attn_drop=0.6) sits between the softmax and the
        # weighted sum: inside the fused kernels where the shape has the cooperative form, else on the
        # composed path (a[E, H] as a tensor, torch's dropout)
        drop = self.attn_drop if (self.training and self.attn_drop.p > 0.0) else None
        fused_drop = None
        if drop is not None and ft.is_cuda and ops.attn_drop_fusable(H, F, ops.DEFAULT_SEG_LEN, get_attention):
            # the mask comes from its own Philox stream (one offset of the generator per call) inside the kernels
            # and is redrawn in the backward: the step stays on the fused path (6.1 -> 1.4 ms per layer step at cfg5)
            from .. import random as _random
            gen = self.noise_generator if self.noise_generator is not None else _random.default_generator
            # the mask's stream is keyed apart from the edge-noise stream: a generator seeded like the one that
            # draws the layer's noise must not hand the mask the (seed, offset) of a noise field
            fused_drop, drop = (float(self.attn_drop.p), gen.seed ^ _random.ATTN_DROP_DOMAIN, gen.next_offset(),
                                gen.device_epoch), None
            if fused_drop[3] is None:
                fused_drop = fused_drop[:3]
        res = ops.gat_aggregate(graph, el, er, ft, self._negative_slope, edge_weight,
                                want_attn=get_attention, attn_fn=drop, attn_drop=fused_drop)
        rst, attn = res if get_attention else (res, None)
        if F_out != F:
            rst, F = rst[..., :F_out], F_out
        if self.res_fc is not None:
            rst = rst + self.res_fc(h).view(h.shape[0], -1, F)
        if self.bias is not None:
            # (the bias gradient of an odd total width goes through ops.column_sum: torch's own column reduction
            # takes a slow path when the width is not a multiple of 4)
            rst = ops.add_bias(rst.reshape(rst.shape[0], H * F), self.bias).view(-1, H, F)
        rst = rst.mean(-2) if self.last else rst.flatten(-2, -1)
        if self.activation:
            rst = self.activation(rst)
        return (rst, attn.unsqueeze(-1)) if get_attention else rst
