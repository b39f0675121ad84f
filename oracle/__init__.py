"""CPU oracle for the realtime-yukarin convert hot path.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the arithmetic of this path lives in third-party packages that are neither
vendored under /root/reference nor installable here (`yukarin`, `become-yukarin` at git HEAD,
un-pinned: /root/reference/requirements.txt:7-8; `chainer` transitively).  The reference's own
tests hold no golden vector for either CNN (/root/reference/tests/test_all_stream.py:21-22,203-213
compares f0 only and needs unshipped models).  This oracle therefore *restates* the published
Chainer operator semantics and the two U-Net topologies and is pinned
only by (i) hand-written known-answer cases and (ii) agreement between three independent
implementations (tap-wise numpy in `ops_numpy`, torch/oneDNN in `torch_ref`, plain-C loop nests in
`ops_ref.c` behind `c_ref`).

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import this
package.  Nothing under `realtime_yukarin_amd/` imports it; the product path has no CPU fallback.
"""
