"""Device-resident transition buffer of the DxMI train step.

The reference keeps transitions in a dict of tensors grown by `torch.cat` (models/DxMI/trainer.py:23-55: T concatenations
per field per append, O(T^2) copies) and reads them with a double fancy index that materialises the permuted T*B buffer
five times per TD step (:278-289).  `TransitionRing` holds the same rows in preallocated HBM:

  traj    [slots, T+1, B, C, H, W]   x_0 .. x_T of every buffered trajectory.  `state` rows are traj[:, :T],
                                     `next_state` rows are traj[:, 1:]: ONE copy of every image serves both fields.
  mean, control [slots, T, B, C, H, W], logp, sigma [slots, T, B], y [slots, B]
  timestep                            not stored: row r of the reference layout has timestep (r mod T*B) div B.

The sampler writes straight into a slot (`VARSampler.sample(..., out=ring.next_slot())`: the fused transition kernel's
outputs ARE the ring rows), so an append moves no data; `gather(key, rows)` is one dxmi_gather_rows launch per field
(INT path).  Row numbering is the reference's: time-major blocks of B per appended trajectory, appends concatenated —
`as_state_dict()` materialises exactly the reference's dict (tests compare the two bit for bit).
"""
import torch

from dxmi_hip import ops

_IMG_KEYS = ("state", "next_state", "mean", "control", "final")


class TransitionRing:
    def __init__(self, n_slots, n_timesteps, batchsize, sample_shape, device, with_y=False, sigma_dims=4):
        self.S, self.T, self.B = n_slots, n_timesteps, batchsize
        self.shape = tuple(sample_shape)
        self.device = torch.device(device)
        T, B = self.T, self.B
        self.traj = torch.empty((n_slots, T + 1, B) + self.shape, dtype=torch.float32, device=device)
        self.mean = torch.empty((n_slots, T, B) + self.shape, dtype=torch.float32, device=device)
        self.control = torch.empty((n_slots, T, B) + self.shape, dtype=torch.float32, device=device)
        self.logp = torch.empty((n_slots, T, B), dtype=torch.float32, device=device)
        self.sigma = torch.empty((n_slots, T, B), dtype=torch.float32, device=device)
        self.y = torch.empty((n_slots, B), dtype=torch.int64, device=device) if with_y else None
        self.sigma_dims = sigma_dims       # VARSampler hands sigma as [B,1,1,1], OpenAIDiffusion as [B]
        self.filled = 0
        self.has = set()

    # ------------------------------------------------------------------ writing
    def next_slot(self):
        """Output views of the next free slot for `sampler.sample(..., out=...)`."""
        if self.filled >= self.S:
            raise RuntimeError(f"TransitionRing: all {self.S} slots are filled (reset_buffer() after update_sampler)")
        s = self.filled
        return {"traj": self.traj[s], "mean": self.mean[s], "control": self.control[s], "logp": self.logp[s],
                "sigma": self.sigma[s], "y": None if self.y is None else self.y[s], "ring": self, "slot": s}

    def append(self, d_sample):
        """trainer.append_buffer semantics.  A sample written in place (it carries `_ring_slot`) is only committed;
        anything else is copied in (one multi-tensor copy per field)."""
        s = self.filled
        if self.filled >= self.S:
            raise RuntimeError("TransitionRing: full")
        x_seq = d_sample["l_sample"]
        assert len(x_seq) == self.T + 1 and x_seq[0].shape == (self.B,) + self.shape, "trajectory shape differs from the ring's"
        in_place = d_sample.get("_ring_slot") == (id(self), s)
        if not in_place:
            torch._foreach_copy_(list(self.traj[s].unbind(0)), [x.detach() for x in x_seq])
        for key, store in (("mean", self.mean), ("control", self.control), ("logp", self.logp), ("sigma", self.sigma)):
            if key in d_sample:
                if not in_place:
                    torch._foreach_copy_(list(store[s].unbind(0)), [v.detach().reshape(store[s][0].shape) for v in d_sample[key][:self.T]])
                self.has.add(key)
        if d_sample.get("y") is not None and self.y is not None:
            if not in_place:
                self.y[s].copy_(d_sample["y"])
            self.has.add("y")
        self.filled += 1
        return self

    def reset(self):
        self.filled = 0
        self.has = set()
        return self

    # ------------------------------------------------------------------ reading (reference row numbering)
    @property
    def n_rows(self):
        return self.filled * self.T * self.B

    def timestep_of(self, rows):
        """INT path: timestep of reference rows (trainer.py:33 `[t] * n_sample` blocks)."""
        return (rows % (self.T * self.B)) // self.B

    def _storage_rows(self, rows, key):
        TB = self.T * self.B
        if key in ("state", "next_state", "final"):
            slot = rows // TB
            r = rows - slot * TB
            base = slot * ((self.T + 1) * self.B)
            if key == "state":
                return base + r
            if key == "next_state":
                return base + r + self.B
            return base + self.T * self.B + r % self.B          # final: x_T of the row's trajectory
        return rows

    def storage_rows(self, rows, key="state"):
        """Row numbers inside the [slots * (T + 1) * B, C, H, W] trajectory block of reference rows `rows` (INT path); the
        next_state of a row is its state row + B."""
        return self._storage_rows(rows, key)

    def gather(self, key, rows):
        """== reference `state_dict[key][rows]` (rows: int64 device tensor of reference row numbers)."""
        rows = rows.contiguous()
        if key == "timestep":
            return self.timestep_of(rows)
        if key in ("state", "next_state", "final"):
            return ops.gather_rows(self.traj.view((-1,) + self.shape), self._storage_rows(rows, key))
        if key in ("mean", "control"):
            return ops.gather_rows(getattr(self, key).view((-1,) + self.shape), rows)
        if key == "logp":
            return ops.gather_rows(self.logp.view(-1), rows)
        if key == "sigma":
            out = ops.gather_rows(self.sigma.view(-1), rows)
            return out.view((-1,) + (1,) * (self.sigma_dims - 1))
        if key == "y":
            TB = self.T * self.B
            return ops.gather_rows(self.y.view(-1), (rows // TB) * self.B + rows % self.B)
        raise KeyError(key)

    def as_state_dict(self):
        """The reference's dict of concatenated tensors (trainer.py:23-55), materialised — for tests and for code written
        against the dict API."""
        n, T, B = self.filled, self.T, self.B
        d = {"state": self.traj[:n, :T].reshape((-1,) + self.shape), "next_state": self.traj[:n, 1:].reshape((-1,) + self.shape),
             "final": self.traj[:n, T:].expand(n, T, B, *self.shape).reshape((-1,) + self.shape),
             "timestep": torch.arange(T, device=self.device).repeat_interleave(B).repeat(n)}
        for key in ("mean", "control"):
            d[key] = getattr(self, key)[:n].reshape((-1,) + self.shape) if key in self.has else torch.FloatTensor().to(self.device)
        d["logp"] = self.logp[:n].reshape(-1) if "logp" in self.has else torch.FloatTensor().to(self.device)
        d["sigma"] = (self.sigma[:n].reshape((-1,) + (1,) * (self.sigma_dims - 1)) if "sigma" in self.has
                      else torch.FloatTensor().to(self.device))
        d["entropy"] = torch.FloatTensor().to(self.device)
        d["y"] = (self.y[:n, None, :].expand(n, T, B).reshape(-1) if "y" in self.has else torch.LongTensor().to(self.device))
        return d

    # dict-style access used by code written against the reference's buffer
    def __getitem__(self, key):
        return self.as_state_dict()[key]


def buffer_rows(state_dict):
    return state_dict.n_rows if isinstance(state_dict, TransitionRing) else state_dict["state"].shape[0]


def buffer_gather(state_dict, key, rows):
    """rows of one buffer field: ring -> gather kernel, reference-style dict -> torch indexing."""
    if isinstance(state_dict, TransitionRing):
        return state_dict.gather(key, rows)
    return state_dict[key][rows]
