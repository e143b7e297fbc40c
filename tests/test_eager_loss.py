"""CPU tests of ``event_plan._EagerLoss`` (``wrap`` needs no GPU): the result of ``contrast_dense`` must behave like the tensor
the autograd engine would have produced -- before, during and AFTER its ``backward()``."""
import pytest
import torch

from event_based_bos_amd.event_plan import _EagerLoss


def make(value=1.5):
    flow = torch.arange(6, dtype=torch.float32).reshape(2, 3).requires_grad_(True)
    d = torch.full((2, 3), 2.0)
    return flow, d.clone(), _EagerLoss.wrap(torch.tensor(value), flow, d)


def test_scaled_relatives_share_one_consumable_gradient():
    """``n = -v; n.backward(); v.backward()``: the engine frees the graph both share, so the second call raises -- it must not hand
    flow.grad a buffer that the first call already scaled in place (silent -2 d before)."""
    flow, d, v = make()
    n = -v
    n.backward()
    assert torch.equal(flow.grad, -d)
    with pytest.raises(RuntimeError, match="second time"):
        v.backward()
    assert torch.equal(flow.grad, -d)  # untouched by the refused call
    with pytest.raises(RuntimeError, match="second time"):
        (v + 0.0).backward()           # attaching after consumption differentiates nothing either
    with pytest.raises(RuntimeError, match="second time"):
        torch.autograd.grad(v, flow)
    # the other order
    flow, d, v = make()
    half = 0.5 * v
    v.backward()
    assert torch.equal(flow.grad, d)
    with pytest.raises(RuntimeError, match="second time"):
        half.backward()
    assert torch.equal(flow.grad, d)


def test_retain_graph_keeps_the_unscaled_gradient_for_relatives():
    flow, d, v = make()
    n = -v
    n.backward(retain_graph=True)
    assert torch.equal(flow.grad, -d)
    v.backward(retain_graph=True)                    # accumulates the UNSCALED gradient: -d + d
    assert torch.equal(flow.grad, torch.zeros_like(d))
    (3.0 * v).backward(gradient=torch.tensor(2.0))   # consumes: 6 d
    assert torch.equal(flow.grad, 6.0 * d)
    with pytest.raises(RuntimeError, match="second time"):
        n.backward()


def test_a_consumed_result_is_still_an_ordinary_tensor():
    """``loss.backward(); print(loss)`` loops: repr / str, cpu, clone, comparisons, stacking and arithmetic for logging all work on
    a consumed result, as they do on any tensor whose graph has been freed."""
    flow, d, loss = make(1.5)
    loss.backward()
    assert "1.5" in repr(loss) and "1.5" in str(loss) and f"{loss:.2f}" == "1.50"
    assert loss.item() == 1.5 and float(loss) == 1.5
    assert loss.cpu().item() == 1.5 and loss.clone().item() == 1.5
    assert bool(loss < 2.0) and bool(loss > 1.0) and bool(loss == 1.5) and not bool(loss != 1.5)
    assert (loss + 1.0).item() == 2.5 and (loss * torch.tensor(2.0)).item() == 3.0 and (-loss).item() == -1.5
    assert torch.stack([loss, loss]).tolist() == [1.5, 1.5]
    assert not bool(torch.isnan(loss))
    running = 0.0
    running += loss.detach()
    assert float(running) == 1.5
    with pytest.raises(RuntimeError, match="second time"):
        loss.backward()


def test_value_reads_before_backward_do_not_attach_a_node():
    flow, d, loss = make(1.5)
    assert "1.5" in repr(loss) and bool(loss < 2.0) and loss.item() == 1.5 and not bool(torch.isnan(loss))
    assert loss._ebos[2] is None                   # still the short cut
    loss.backward()
    assert loss._ebos[2] is None and torch.equal(flow.grad, d)
