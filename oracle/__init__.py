"""CPU oracle for the GAN_SR_wind_field 3D-conv GAN train-step hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``gan_sr_wind_field_amd/`` may import
this package: it exists so that ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` can check (never replace) the HIP path.

It is a from-scratch, *functional* restatement (plain ``torch.nn.functional``
calls over a flat ``{state_dict key: tensor}`` mapping, fp32/fp64 on the CPU)
of the reference algorithm; every function cites the reference ``file:line``
it follows.  Parity pinning: ``tests/golden/*.npz`` were produced by importing
the real reference from ``/root/reference`` (``tests/golden/make_golden.py``)
and ``tests/test_oracle_golden.py`` checks this oracle against every one of
them, so the oracle is *pinned*, not "parity unpinned".
"""
