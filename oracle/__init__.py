"""oracle/ -- TEST INFRASTRUCTURE ONLY.

A CPU restatement of the OpenTAL/AFSD detection hot path (reference:
Cogito2012/OpenTAL).  It exists to *check* the HIP product path; it is never the
thing that is shipped or measured.  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import anything from here.
``opental_amd/`` must never import ``oracle``.

Pinning status (see DESIGN.md "Oracle"):
  * The reference repository has NO tests and NO golden vectors for this path
    (SURVEY.md section 4), so nothing of the reference's own pins it.
  * The Python part of the reference imports on CPU in the build container; the
    restatement here was checked against it there by ``oracle/pin_against_reference.py``
    and the resulting input/output vectors are committed under ``tests/golden/``.
  * The reference's only native op (BoundaryMaxPooling, CUDA + THC headers) cannot
    be compiled here (no nvcc, THC removed from torch).  ``oracle/bmp_ref.c`` restates
    ``AFSD/prop_pooling/boundary_max_pooling_kernel.cu:17-82`` and is pinned by
    hand-computed known answers + a brute-force Python loop.  For that op alone:
    *parity unpinned by the reference*.
"""
