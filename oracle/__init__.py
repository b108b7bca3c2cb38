"""CPU oracle for the Howl audio hot path -- TEST INFRASTRUCTURE ONLY.

This package is a from-scratch CPU restatement (torch-CPU / numpy) of the
reference algorithm for the path

    16 kHz PCM -> STFT -> mel filterbank -> log (+deltas, ZMUV)
               -> res8 / LSTM classifier forward + backward -> loss -> AdamW

Every function cites the reference file:line it follows (paths relative to
the castorini/howl checkout).  It exists so that the hand-written HIP path in
``howl_amd`` can be checked against something that is known to agree with the
reference:

* the oracle itself is pinned against golden vectors captured by importing the
  reference's own modules (``tests/golden/make_golden.py``; fixtures committed
  under ``tests/golden/*.npz``) -- see ``tests/test_oracle_golden.py``;
* the HIP path is then compared with the oracle on seeded inputs
  (``tests/test_gpu_*.py``).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this package.  Nothing under ``howl_amd``
imports it; the product path raises if the HIP extension is missing.

Third-party arithmetic restated here (absent from the reference checkout):
torchaudio 0.10.1 ``MelSpectrogram`` / ``ComputeDeltas`` (pinned by
``requirements.txt:15-16`` as ``torchaudio>=0.5.0`` next to ``torch==1.10.1``).
"""
