"""
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Plain-PyTorch CPU restatement of the Graphormer layer used by GHN-3.
Follows /root/reference/ghn3/graphormer.py:
  * layer-0 prologue (centrality / input-distance embeddings, node mask, fw/bw edge stack)
        graphormer.py:219-237
  * edge embedding + projection to per-head attention bias      graphormer.py:49-68,94-99,114-117
  * multi-head self-attention with additive edge bias and -2**15 mask fill
        graphormer.py:119-142
  * pre-LN residual block with erf-GELU feed-forward            graphormer.py:22-47,239-241

Pinned by tests/test_oracle.py against golden vectors produced by importing the reference's
graphormer.py UNMODIFIED (tests/golden/make_golden.py).

All functions take a flat ``{name: tensor}`` parameter dict ``p`` and a key prefix, using the
reference's state-dict names (SURVEY 8(a) "State-dict layout").
"""

import torch
import torch.nn.functional as F

MAX_DEGREE = 100        # graphormer.py:196
MAX_INPUT_DIST = 1000   # graphormer.py:197
MASK_FILL = -2.0 ** 15  # graphormer.py:135


def layer0_prologue(x, edges, mask, p, pre):
    """graphormer.py:229-237.  x (B,N,C) float, edges (B,N,N) int64, mask (B,N,N) bool or None."""
    one_hop = (edges == 1).long()
    deg_in = torch.clip(one_hop.sum(1), 0, MAX_DEGREE)          # column sums  (B,N)
    deg_out = torch.clip(one_hop.sum(2), 0, MAX_DEGREE)         # row sums     (B,N)
    dist0 = torch.clip(edges[:, 0, :], 0, MAX_INPUT_DIST)       # distance from node 0 (B,N)
    x = x + p[pre + 'centrality_embed_in.weight'][deg_in]
    x = x + p[pre + 'centrality_embed_out.weight'][deg_out]
    x = x + p[pre + 'input_dist_embed.weight'][dist0]
    if mask is not None:
        x = x * mask[:, :, :1]
    edges2 = torch.stack((edges, edges.permute(0, 2, 1)), dim=-1) + 2
    return x, edges2


def edge_bias(edges2, p, pre):
    """graphormer.py:114-117.  edges2 (B,N,N,2) int64 -> (B,N,N,H) float."""
    e = p[pre + 'attn.edge_embed.embed.weight'][edges2]             # (B,N,N,2,C)
    e = e.reshape(*e.shape[:-2], -1)                                # (B,N,N,2C)
    h = F.relu(F.linear(e, p[pre + 'attn.proj_e.0.weight'], p[pre + 'attn.proj_e.0.bias']))
    return F.linear(h, p[pre + 'attn.proj_e.2.weight'], p[pre + 'attn.proj_e.2.bias'])


def attention(x, bias, mask, p, pre, heads, return_probs=False):
    """graphormer.py:119-142.  x is LN1(x) (B,N,C); bias (B,N,N,H) or None; mask (B,N,N) bool or None."""
    B, N, C = x.shape
    d = C // heads
    qkv = F.linear(x, p[pre + 'attn.to_qkv.weight']).reshape(B, N, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.unbind(0)
    attn = (q @ k.transpose(-2, -1)) * (d ** -0.5)
    if bias is not None:
        attn = attn + bias.permute(0, 3, 1, 2)
    if mask is not None:
        attn = attn.masked_fill(~mask.unsqueeze(1), MASK_FILL)
    attn = attn.softmax(dim=-1)
    o = (attn @ v).transpose(1, 2).reshape(B, N, C)
    o = F.linear(o, p[pre + 'attn.to_out.0.weight'], p[pre + 'attn.to_out.0.bias'])
    return (o, attn) if return_probs else o


def feed_forward(x, p, pre):
    """graphormer.py:38-47 with act_layer = nn.GELU (exact erf form)."""
    h = F.gelu(F.linear(x, p[pre + 'ff.net.0.weight'], p[pre + 'ff.net.0.bias']))
    return F.linear(h, p[pre + 'ff.net.3.weight'], p[pre + 'ff.net.3.bias'])


def transformer_layer(x, edges, mask, p, pre, heads, layer0, eps=1e-5):
    """
    graphormer.py:208-248 for a (B,N,C) input.
    layer0: ``edges`` is the (B,N,N) int64 shortest-path matrix; otherwise it is the (B,N,N,H)
    bias returned by layer 0 (re-added in every layer, graphormer.py:126-130) or None.
    Returns (x, bias).
    """
    C = x.shape[-1]
    if layer0:
        x, edges2 = layer0_prologue(x, edges, mask, p, pre)
        bias = edge_bias(edges2, p, pre)
    else:
        bias = edges
    h = F.layer_norm(x, (C,), p[pre + 'ln1.weight'], p[pre + 'ln1.bias'], eps)
    x = x + attention(h, bias, mask, p, pre, heads)
    h = F.layer_norm(x, (C,), p[pre + 'ln2.weight'], p[pre + 'ln2.bias'], eps)
    x = x + feed_forward(h, p, pre)
    return x, bias
