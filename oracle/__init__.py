"""CPU ORACLE - test infrastructure, NOT product code.

Float64 NumPy / SciPy restatements of the reference algorithm for the SparseVFC hot path:

* ``sparsevfc_oracle``  dynamo's ``SparseVFC`` / ``get_P`` / ``lstsq_solver`` / ``bandwidth_selector`` / ``con_K``, op for op
                        (call sites: /root/reference/spateo/tdr/morphometrics/morphofield/sparsevfc.py:167,189-198,234);
* ``streamed_oracle``   the same EM with the N x M kernel matrix generated chunk by chunk, for the sizes the in-memory form
                        cannot hold (2 M x 2000 ... 8 M x 3000); bit-identical to the in-memory oracle with chunked sums;
* ``dg_oracle``         Jacobian + differential-geometry evaluators (twins: .../morphofield_dg/GPVectorField.py:12-190);
* ``align_oracle``      the alignment module's con_K / BA_transform / _update_nonrigid (alignment/methods/morpho_class.py);
* ``trajectory_oracle`` dynamo ``fate``'s arc-length sampling as driven by morphopath (trajectory.py:61-115).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` / ``parity`` legs may import this package;
the product (``spateo-release_amd/``) never does and has no CPU fallback.  PARITY STATUS: the in-tree twins are pinned by
goldens produced by the real reference code (tests/golden/); everything that lives only in dynamo-release (not vendored in
/root/reference, not installable here) is **parity unpinned** - see each module's header.
"""
