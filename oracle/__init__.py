"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference hot path.

Nothing under ``oracle/`` is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import, link or execute anything from this directory, and only as the checker
(or as the timed CPU baseline), never as the thing shipped.  The product path
(``chiron_amd``) runs exclusively through the HIP library
``chiron_amd/csrc/libchiron_amd.so`` and fails loudly when it is missing.

Parity status (see DESIGN.md "Oracle"):
  * host stages (windowing, batch packing, sparse slicing, assembly, qs):
    PINNED by golden vectors captured from the reference's own Python
    functions (tests/golden/make_golden.py) and by the reference's checked-in
    example outputs (segments -> consensus, exact for all 5 reads).
  * NN stages (CNN / BiLSTM / FC) and the two TF CTC decoders: the arithmetic
    lives in tensorflow==1.15.0 (setup.py:28-29), which is absent from
    /root/reference and not installable here; the trained weights
    (*.data-00000-of-00001) are stripped too.  These restatements follow the
    op composition recorded in the shipped .meta graphs and the published TF
    1.15 kernels.  PARITY UNPINNED against real TF outputs; cross-checked
    against torch CPU (conv1d / LSTMCell) and brute-force CTC enumeration.
"""
