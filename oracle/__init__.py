"""CPU oracle for the tag-aware-recommendation hot path.

TEST INFRASTRUCTURE ONLY.  This package restates, on the CPU (numpy + PyTorch
CPU ops), what the reference's hot path computes, so that the HIP kernels can
be checked against it on a GPU box where `/root/reference` does not exist.

Who may import it: `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline`
leg of `bench.py` -- as the checker / the timed CPU baseline, never as the
thing shipped.  Nothing under `tag-aware-recommendation_amd/` imports it.

Parity status: PINNED.  Every function here is checked, in this container,
against the reference itself imported unmodified from `/root/reference`
(`oracle/make_golden.py`, three in-process shims); the
captured inputs/outputs are committed under `tests/golden/*.npz` and
`tests/test_oracle_golden.py` re-checks the oracle against them without the
reference.
"""
