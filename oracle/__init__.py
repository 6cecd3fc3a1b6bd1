"""CPU oracle for the orcAI spectrogram -> label hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is imported by the product
package ``orcai_amd``; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and there only as the
checker.  The product path fails loudly when the HIP library is missing.

Parity pinning ("Oracle"):

* ``postprocess_ref`` and ``frontend_ref.preprocess_spectrogram_ref`` are pinned
  bit-exactly by golden vectors produced by the reference's own numpy/pandas
  functions (``tests/golden/make_golden.py`` imports them from
  ``/root/reference`` with the absent third-party modules stubbed).
* ``frontend_ref.stft_ref`` / ``amplitude_to_db_ref`` restate librosa 0.11.0
  (absent from this image) from its documented semantics: **parity unpinned**
  at the librosa boundary.
* ``model_ref`` restates Keras 3.10 / TensorFlow 2.19 layer semantics (absent):
  **parity unpinned** at the Keras boundary; it is cross-checked by an
  independent explicit-loop numpy implementation (``model_ref_loops``).
"""
