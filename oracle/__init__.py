"""CPU oracle for the pytorch_sound hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``pytorch_sound_amd/`` may import this package.  The only legal
importers are ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` - and there only as the checker / reported baseline, never
as the thing shipped or measured as the product.

Pinning status (see DESIGN.md "Oracle"):
  * STFT / framing / log-mel arithmetic / attention blocks / HiFi-GAN / Trainer /
    registry: pinned against golden vectors produced by importing the reference
    (``/root/reference``) in the build container - ``tools/gen_golden.py`` ->
    ``tests/golden/*.npz``.
  * mel filterbank VALUES (``librosa.filters.mel``, librosa==0.8.0, absent from the
    reference tree and from this image): restated from the published algorithm,
    "parity unpinned" for the values; known-answer tests stand in
    (tests/test_oracle_mel.py).
"""
