"""Training driver — the CALLER of the hot path; mirrors mentflow/train/train.py:18-283 (penalty-method outer loop,
AdamW inner loop, NaN guard, best-state tracking) without the plotting/pickle side effects.

Differences, all on the host side: the per-iteration scalars (L, H, mean D) are fetched with ONE device->host copy
instead of the reference's six synchronising conversions per iteration (train.py:167-214), and the history is kept in
memory (``self.history``) instead of being re-pickled every iteration (utils/logging.py:64-66).

``graphed=True`` replays the whole iteration (zero_grad + loss + backward + optimizer step) from a hipGraph captured
once per epoch (the penalty parameter is a capture-time constant; the learning rate too: a scheduler change triggers a
re-capture) — the launch-bound small-batch regime of the reference (25 000 particles).  Needs a capturable optimizer
(``torch.optim.AdamW(..., capturable=True)``).  A non-finite loss is handled as the reference does: that iteration's
update is undone (mentflow_amd.graph.GraphedTrainStep.undo_last_step)."""
from __future__ import annotations

import copy
import time
from typing import Callable, Dict, List, Optional

import torch


class Trainer:
    def __init__(self, model, optimizer, lr_scheduler=None, plot: Optional[Callable] = None, eval: Optional[Callable] = None,
                 output_dir: Optional[str] = None, notebook: bool = False, load_best: bool = True, verbose: bool = True,
                 graphed: bool = False, lr_on_device: bool = False) -> None:
        """graphed: replay every epoch's steps from a hipGraph.  A float learning rate is baked into the graph: when a
        scheduler moves it the epoch's graph is re-captured (3 eager warm-up steps + capture) — bit-identical to the eager
        Trainer.  lr_on_device=True keeps the rate in a device tensor per parameter group instead (no re-capture at all:
        for schedulers that move the rate often); torch's AdamW then multiplies by an fp32 tensor where it folds a python
        float otherwise, so results differ from the float-rate run in the last bits."""
        self.model = model
        self.optimizer = optimizer
        self.lr_scheduler = lr_scheduler
        self.plot = plot
        self.eval = eval
        self.output_dir = output_dir
        self.load_best = load_best
        self.verbose = verbose
        self.graphed = graphed
        self.lr_on_device = lr_on_device
        self.history: Dict[str, List] = {}

    def _log(self, info: dict) -> None:
        for k, v in info.items():
            self.history.setdefault(k, []).append(v)

    def train(self, epochs: int = 20, iterations: int = 1000, batch_size: int = 30000, rtol: float = 0.05, atol: float = 0.0,
              dmax: float = 0.0, penalty_start: float = 0.0, penalty_step: float = 20.0, penalty_scale: float = 1.1,
              penalty_max: Optional[float] = None, eval_freq: Optional[int] = None, eval_batch_size: int = 100000,
              **_ignored) -> None:
        if penalty_max is None:
            penalty_max = float("inf")
        if not eval_freq:
            eval_freq = iterations
        start_time = time.time()
        model = self.model

        def train_epoch(epoch):
            best_loss = float("inf")
            best_state_dict = copy.deepcopy(model.state_dict())
            gstep = None
            lr_dev = None
            if self.graphed:
                from .graph import GraphedTrainStep
                if self.lr_on_device:
                    # the learning rate lives in a DEVICE tensor per parameter group (capturable optimizers read it inside
                    # the graph), so a scheduler that moves the rate needs no re-capture: the new value is written into
                    # the same tensor.  Schedulers assign a python float to group["lr"]; it is moved back below.
                    dev = next(model.parameters()).device
                    lr_dev = []
                    for g in self.optimizer.param_groups:
                        t = g["lr"] if torch.is_tensor(g["lr"]) else torch.tensor(float(g["lr"]), dtype=torch.float32, device=dev)
                        g["lr"] = t
                        lr_dev.append(t)
                gstep = GraphedTrainStep(model, self.optimizer, batch_size, guard=True)
                lr_captured = [g["lr"] for g in self.optimizer.param_groups]
            for iteration in range(iterations):
                if gstep is not None:
                    if lr_dev is not None:
                        for g, t in zip(self.optimizer.param_groups, lr_dev):
                            if g["lr"] is not t:                                           # the scheduler assigned a new float
                                t.fill_(float(g["lr"]))
                                g["lr"] = t
                    elif [g["lr"] for g in self.optimizer.param_groups] != lr_captured:    # float rate: baked into the graph
                        gstep.recapture()
                        lr_captured = [g["lr"] for g in self.optimizer.param_groups]
                    gstep.step()
                    L_val, H_val, D_val = (float(v) for v in gstep.scalars().cpu())         # one sync
                    if L_val != L_val or L_val in (float("inf"), float("-inf")):            # train.py:167
                        gstep.undo_last_step()
                else:
                    self.optimizer.zero_grad()
                    loss, H, D = model.loss(batch_size)
                    scalars = torch.stack([loss.detach(), H.detach() if torch.is_tensor(H) else torch.tensor(float(H), device=loss.device),
                                           torch.stack([d.detach() for d in D]).mean()]).cpu()     # one sync
                    L_val, H_val, D_val = (float(v) for v in scalars)
                    if not (L_val != L_val or L_val in (float("inf"), float("-inf"))):              # train.py:167
                        loss.backward()
                        self.optimizer.step()
                self._log(dict(epoch=epoch, iteration=iteration, L=L_val, H=H_val, D_norm=D_val, batch_size=batch_size,
                               # (a device-tensor rate is read back here: schedulers such as ReduceLROnPlateau update it IN PLACE,
                               # so no host copy can follow it; the stream is idle behind the scalar fetch above)
                               learning_rate=float(self.optimizer.param_groups[0]["lr"]), penalty=model.penalty_parameter,
                               time=time.time() - start_time))
                if L_val < best_loss:                                                             # train.py:197-199
                    best_loss = L_val
                    best_state_dict = copy.deepcopy(model.state_dict())
                if ((iteration + 1) % eval_freq == 0) or ((iteration + 1) == iterations):         # train.py:202-211
                    if self.eval is not None or self.plot is not None:
                        model.eval()
                        with torch.no_grad():
                            curr = copy.deepcopy(model.state_dict())
                            if self.load_best:
                                model.load_state_dict(best_state_dict)
                            if self.eval is not None:
                                self.eval(model)
                            if self.plot is not None:
                                self.plot(model)
                            model.load_state_dict(curr)
                        model.train()
                if self.lr_scheduler is not None:
                    self.lr_scheduler.step(L_val)                                                 # train.py:214
            if lr_dev is not None:
                # hand the optimizer back with plain float rates: state_dict() / checkpoints and a later non-capturable use must
                # not find device tensors in param_groups (one read-back per epoch)
                for g in self.optimizer.param_groups:
                    g["lr"] = float(g["lr"])
            return best_state_dict

        converged, final_epoch = False, False
        D_norm_old = float("inf")
        model.penalty_parameter = penalty_start
        best_state_dict = copy.deepcopy(model.state_dict())
        for epoch in range(epochs):
            if self.verbose:
                print(f"epoch = {epoch}\npenalty = {model.penalty_parameter}")
            best_state_dict = train_epoch(epoch)
            with torch.no_grad():                                                                  # train.py:235-244
                model.eval()
                current = copy.deepcopy(model.state_dict())
                model.load_state_dict(best_state_dict)
                _, _, D = model.loss(batch_size=eval_batch_size)
                D_norm = float(torch.stack([d.detach() for d in D]).mean())
                model.load_state_dict(current)
                model.train()
            if self.verbose:
                print("D_norm = {:0.3e}\nD_norm_old = {:0.3e}".format(D_norm, D_norm_old))
            converged_message = ""
            if D_norm <= dmax:
                converged, converged_message = True, "CONVERGED (dmax)"
            if D_norm > (1.0 - rtol) * D_norm_old:
                converged, converged_message = True, "CONVERGED (rtol)"
            if D_norm_old - D_norm < atol:
                converged, converged_message = True, "CONVERGED (atol)"
            if converged:
                if final_epoch:
                    model.load_state_dict(best_state_dict)
                    return
                if self.verbose:
                    print(converged_message + "\nTraining one more epoch with same penalty parameter")
            else:
                model.penalty_parameter *= penalty_scale
                model.penalty_parameter += penalty_step
                if model.penalty_parameter >= penalty_max:
                    if self.verbose:
                        print("Max penalty parameter reached.")
                    return
            final_epoch = converged
            D_norm_old = D_norm
        model.load_state_dict(best_state_dict)
