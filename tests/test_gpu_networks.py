"""GPU tests of SURVEY 8(f) row 2: NetworkLight target networks whose parameters the HIP GHN predicts, run on images,
with the loss back-propagated into the GHN -- the step train_ghn_ddp.py performs (trainer.py:282-351)."""

import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import recipe
from util_parity import make_models

pytestmark = pytest.mark.gpu


def _oracle_batch(gb):
    from oracle import ghn3_ref as R
    return R.GraphBatchRef([R.GraphRef(nf, ni, A) for nf, ni, A in zip(gb.node_feat, gb.node_info, gb.edges)])


def test_light_networks_trained_through_the_ghn_vs_oracle():
    """Two sampled architectures (light layers): the HIP GHN assigns every tensor of their parameter tables, the
    networks run on an image batch, cross-entropy + auxiliary-free logits; loss and GHN gradients against the CPU
    oracle GHN driving the same light networks."""
    from ghn3_amd.deepnets1m import SampledNets
    hip, oracle = make_models(recipe.TINY_CFG, recipe.TINY_SEED)
    hip.train()
    oracle.train()
    gen = torch.Generator().manual_seed(3)
    images = torch.randn(4, 3, 32, 32, generator=gen)
    labels = torch.tensor([2, 0, 9, 4])

    gb = next(SampledNets.loader(meta_batch_size=2, seed=5, max_nodes=120))
    gb_o = next(SampledNets.loader(meta_batch_size=2, seed=5, max_nodes=120))
    assert [len(c) for c in gb.nets[0]._layered_modules] == [len(c) for c in gb_o.nets[0]._layered_modules]

    nets = hip(gb.nets, gb.to_device('cuda'), keep_grads=True)
    loss = 0.
    for net in nets:
        for cell in net._layered_modules:
            for name, e in cell.items():
                t = getattr(e['module'], 'weight' if e['is_w'] else 'bias')
                assert isinstance(t, torch.Tensor) and t.is_cuda and tuple(t.shape) == e['sz'], name
        net.eval()                      # (Dropout of a two-layer head off; light BatchNorm keeps batch statistics)
        logits, aux = net(images.cuda())
        assert aux is None and logits.shape == (4, 10)
        loss = loss + F.cross_entropy(logits, labels.cuda())
    loss.backward()
    torch.cuda.synchronize()

    nets_o, _ = oracle(gb_o.nets, _oracle_batch(gb_o), keep_grads=True)
    for net, net_o in zip(nets, nets_o):
        net_o.eval()
        worst = 0.0
        for cell, cell_o in zip(net._layered_modules, net_o._layered_modules):
            for name, e in cell.items():
                attr = 'weight' if e['is_w'] else 'bias'
                t, t_o = getattr(e['module'], attr), getattr(cell_o[name]['module'], attr)
                err = float((t.detach().cpu().double() - t_o.detach().double()).norm() / (t_o.detach().double().norm() + 1e-12))
                worst = max(worst, err)
                assert err < 1e-4, (name, tuple(t_o.shape), err)
    loss_o = sum(F.cross_entropy(net(images)[0], labels) for net in nets_o)
    loss_o.backward()
    assert abs(loss.item() - loss_o.item()) < 2e-4 * max(1.0, abs(loss_o.item())), (loss.item(), loss_o.item())
    po = dict(oracle.named_parameters())
    seen = 0
    for k, p in hip.named_parameters():
        go = po[k].grad
        if go is None or float(go.norm()) < 1e-7:
            continue
        assert p.grad is not None, k
        err = float((p.grad.cpu().double() - go.double()).norm())
        assert err < 2e-3 * float(go.norm()) + 1e-6, (k, err, float(go.norm()))
        seen += 1
    assert seen > 20


def test_trainer_steps_on_sampled_light_networks():
    """Trainer.update on the architecture stream: three optimizer steps on light networks (cross-entropy on images +
    predicted-parameter regulariser), finite metrics, parameters move, nothing skipped."""
    from ghn3_amd import Trainer
    from ghn3_amd.deepnets1m import SampledNets
    hip, _ = make_models(recipe.TINY_CFG, recipe.TINY_SEED)
    gen = torch.Generator().manual_seed(1)
    images = torch.randn(4, 3, 32, 32, generator=gen)
    targets = torch.tensor([1, 7, 3, 9])
    tr = Trainer(hip, 'adamw', {'lr': 1e-3, 'weight_decay': 1e-2}, 'cosine', n_batches=3, grad_clip=5, device='cuda',
                 log_interval=1, predparam_wd=3e-5, epochs=2)
    queue = SampledNets.loader(meta_batch_size=2, seed=9, max_nodes=120)
    before = hip._flat.detach().clone()
    for step in range(3):
        m = tr.update(images, targets, graphs=next(queue))
        tr.log(step)
    avg = m.avg()
    assert np.isfinite(avg['loss']) and 0.0 <= avg['top1'] <= 100.0 and tr.skipped_updates == 0
    assert not torch.equal(before, hip._flat)


def test_train_ghn_ddp_counterpart_script(tmp_path):
    """examples/train_ghn_ddp.py (the train_ghn_ddp.py:87-150 sequence on the sampled architecture stream): a few steps
    with checkpoints, then a second run that resumes from the checkpoint the first one wrote."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, 'examples', 'train_ghn_ddp.py'), '--steps', '5', '--meta-batch-size', '2',
           '--batch-size', '8', '--workers', '2', '--max-nodes', '120', '--save', str(tmp_path)]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:]
    assert 'ms per step' in out.stdout and os.path.exists(tmp_path / 'checkpoint.pt'), out.stdout[-1500:]
    out2 = subprocess.run(cmd + ['--epochs', '2'], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert out2.returncode == 0, out2.stdout[-3000:]
    assert 'resuming' in out2.stdout, out2.stdout[-1500:]
