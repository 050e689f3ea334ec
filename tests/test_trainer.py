"""The caller of the hot path: mentflow_amd.train.Trainer (mirror of mentflow/train/train.py) drives MENTFlow.loss,
AdamW and the penalty schedule; checkpoints round-trip.  Runs on the emulated kernels (tiny sizes) and on the GPU."""
import torch

import mentflow_amd as mf
from mentflow_amd.harness import build_problem


def test_trainer_reduces_loss_and_checkpoint_roundtrip(backend, tmp_path):
    prob = build_problem(ndim=2, num=3, bins=16, xmax=3.5, seed=21, transforms=2, prior_scale=1.0, device=backend,
                         meas_samples=4000, dist_name="swissroll", optics="2d_linear")
    model = prob.model
    torch.manual_seed(0)
    opt = torch.optim.AdamW(model.parameters(), lr=5e-3, weight_decay=0.0)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, min_lr=5e-3, patience=400, factor=0.1)
    trainer = mf.train.Trainer(model, opt, sched, verbose=False)
    trainer.train(epochs=2, iterations=6, batch_size=256, rtol=-1, atol=-1, dmax=1e-9, penalty_start=20.0,
                  penalty_step=25.0, penalty_scale=1.25, eval_batch_size=512)
    hist = trainer.history
    assert len(hist["L"]) == 12 and hist["penalty"][0] == 20.0 and hist["penalty"][-1] == 20.0 * 1.25 + 25.0
    assert all(v == v for v in hist["L"])                     # no NaN
    assert min(hist["D_norm"][6:]) < hist["D_norm"][0]        # the data mismatch goes down
    path = tmp_path / "model.pt"
    model.save(str(path))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        for p in model.parameters():
            p.add_(1.0)
    model.load(str(path), device=backend)
    for k, v in model.state_dict().items():
        assert torch.equal(v, sd[k])
    assert all(k.startswith("generator.") for k in sd)        # transforms/diagnostics are not in state_dict (core.py)
    x = model.sample(64)
    assert x.shape == (64, 2)
