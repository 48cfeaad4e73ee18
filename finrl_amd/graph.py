"""hipGraph capture of launch-bound rollout segments.

Every ``step()`` of the batched envs is a plain kernel launch on the caller's stream (no
allocation, no synchronisation, no host read-back), so a whole segment -- policy forward, action
copy, env step, writes into the rollout buffer -- can be captured once with
``torch.cuda.CUDAGraph`` (a hipGraph on ROCm) and replayed with one host call.  At small batch
sizes the eager path is bound by the ~13 us a Python/ctypes launch costs, not by the GPU.
"""
from __future__ import annotations


class GraphedSegment:
    """Capture ``n_steps`` of ``buf.collect``-style stepping into one graph.

    ``policy(obs) -> (actions, values, log_probs)`` must consist of capturable torch ops on
    device tensors (no host sync, no data-dependent Python control flow).  ``replay()`` runs the
    segment from the env's current state and returns the buffer; ``buf.obs[0]`` must hold the
    observation to start from (``replay`` copies ``first_obs`` there when given).
    """

    def __init__(self, env, policy, buf, warmup=2):
        import torch
        self.env, self.buf = env, buf
        dev = buf.obs.device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        snapshot = self._snapshot()
        with torch.cuda.stream(side):              # warm-up launches outside the capture
            for _ in range(warmup):
                self._segment(policy)
        torch.cuda.current_stream(dev).wait_stream(side)
        self._restore(snapshot)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._segment(policy)
        self._restore(snapshot)                    # capture does not execute; warm-up did

    def _snapshot(self):
        return {k: v.clone() for k, v in self._state_tensors().items()}

    def _restore(self, snap):
        for k, v in self._state_tensors().items():
            v.copy_(snap[k])

    def _state_tensors(self):
        env = self.env
        out = {}
        for name in ("_f64", "_i32", "_f32", "_state_f64", "_state_i32", "_stocks"):
            t = getattr(env, name, None)
            if t is not None:
                out[name] = t
        return out

    def _segment(self, policy):
        buf, env = self.buf, self.env
        for t in range(buf.n_steps):
            a, v, lp = policy(buf.obs[t])
            buf.step(env, t, a, v, lp)

    def replay(self, first_obs=None):
        if first_obs is not None:
            self.buf.obs[0].copy_(first_obs)
        self.graph.replay()
        return self.buf
