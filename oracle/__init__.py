"""CPU oracle for the ntm-tracker hot path.  TEST INFRASTRUCTURE ONLY.

Everything under ``oracle/`` is a CPU restatement of the reference's
arithmetic (JeffOwOSun/ntm-tracker; citations are ``file:line`` into the
reference tree).  It exists to *check* the HIP path.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it; the product package (``ntmtrack`` / ``ntm-tracker_amd``) never does.

Pinning status (see DESIGN.md "Oracle"):
  * ``dnc_oracle``   -- pinned by the reference's own DNC tests' planted /
    known-answer cases (dnc/addressing_test.py, dnc/access_test.py,
    dnc/util_test.py), restated in tests/test_oracle_dnc.py.
  * ``ntm_oracle.batched_smooth_cosine_similarity`` -- the reference's only
    NTM test (ops_test.py:20-34) pins the *intended* smooth-cosine semantics,
    which the shipped code (ops.py:147-156) does not compute; both modes are
    implemented, the test's vector pins ``mode="smooth_cosine"`` and the
    hand-evaluated as-coded values pin the default.
  * NTMCell step, circular convolution, serialiser, VGG stack, loss,
    RMSProp/clip, both LSTM cells: **parity unpinned** -- no reference test,
    fixture or runnable reference (TF1/Sonnet/Py2 are absent from the image)
    covers them; they follow the cited lines and the documented formulas of
    the un-vendored third-party ops (TensorFlow 1.x, dm-sonnet v1, versions
    unpinned by the reference).
"""
