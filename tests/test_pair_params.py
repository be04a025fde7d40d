"""training.pair_params / pair_storage (round 4): two parameters of the paired backbones re-homed side by side, seen as one [2, ...] tensor,
gradients handed back as views.  Host logic: runs on CPU."""
import torch

from keypointfusion_amd import training as T


def test_pair_params_rehomes_storage_and_splits_the_gradient():
    a, b = torch.nn.Parameter(torch.arange(12.0).view(3, 4)), torch.nn.Parameter(-torch.arange(12.0).view(3, 4))
    a0, b0 = a.detach().clone(), b.detach().clone()
    opt = torch.optim.SGD([a, b], lr=0.5)
    reg = {}
    both = T.pair_params(reg, "w", a, b)
    assert both.shape == (2, 3, 4) and torch.equal(both[0], a0) and torch.equal(both[1], b0)
    assert a.data_ptr() == reg["w"].data_ptr() and b.data_ptr() == reg["w"].data_ptr() + 12 * 4, "the two parameters are views of one allocation"
    assert torch.equal(a, a0) and torch.equal(b, b0), "values preserved"
    w = torch.arange(24.0).view(2, 3, 4)
    (both * w).sum().backward()
    assert torch.equal(a.grad, w[0]) and torch.equal(b.grad, w[1])
    assert a.grad.data_ptr() + 48 == b.grad.data_ptr(), "the halves of one gradient tensor were adopted as views (no copies)"
    opt.step()  # the optimiser keeps working on the re-homed parameters
    assert torch.equal(a, a0 - 0.5 * w[0]) and torch.equal(reg["w"][1], b0 - 0.5 * w[1])
    again = T.pair_params(reg, "w", a, b)
    assert again.data_ptr() == both.data_ptr(), "a second forward finds the pairing in place"
    a.data = a.data.clone()  # something moved a parameter (module.to(), load with assign=True): paired again, values kept
    moved = T.pair_params(reg, "w", a, b)
    assert moved.data_ptr() != both.data_ptr() and torch.equal(moved[0], a0 - 0.5 * w[0]) and a.data_ptr() == moved.data_ptr()


def test_pair_storage_of_buffers_and_odd_sizes():
    rm, rv = torch.zeros(8), torch.ones(8)
    reg = {}
    both = T.pair_storage(reg, "running", rm, rv)
    both.view(-1)[:] = torch.arange(16.0)
    assert torch.equal(rm, torch.arange(8.0)) and torch.equal(rv, torch.arange(8.0, 16.0)), "in-place updates of the pair reach both buffers"
    n0, n1 = torch.tensor(3), torch.tensor(4)
    assert T.pair_storage(reg, "nbt", n0, n1) is None, "sizes that would misalign the second half are not paired"
    p, q = torch.nn.Parameter(torch.ones(3)), torch.nn.Parameter(torch.zeros(3))
    t = T.pair_params(reg, "odd", p, q)  # falls back to a real stack, still differentiable
    t.sum().backward()
    assert t.shape == (2, 3) and torch.equal(p.grad, torch.ones(3)) and torch.equal(q.grad, torch.ones(3))
