"""CPU oracle of the wavehip hot path -- TEST INFRASTRUCTURE ONLY.

numpy restatements (`ref_np.py`) and plain-C restatements (`c4fm_ref.c`, `cqpsk_ref.c`, `lsm_ref.c` with their
ctypes wrappers) of the reference algorithms, pinned by the golden vectors `gen_golden.py` captures from the real
reference.  Only tests/, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg import this package; the product
(`wavecap-sdr_amd/wavehip`) never does (tests/test_abi.py checks)."""
