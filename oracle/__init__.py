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
  * NN stages (CNN / BiLSTM / FC): the arithmetic lives in tensorflow==1.15.0
    (setup.py:28-29), absent from /root/reference and not installable here, and
    the trained weights (*.data-00000-of-00001) are stripped, so there are no
    TF-produced logits.  The COMPOSITION is pinned to the reference all the
    same: tests/golden/make_meta_golden.py executes the node lists of the
    shipped MetaGraphDefs (chiron/model/*/final.ckpt-*.meta) op by op in
    float64 and tests/test_meta_golden.py holds nn_oracle to those activations
    at 1e-12, plus a structural digest (strides, BN epsilon / association,
    LSTM gate order, forget bias, masking, ReverseSequence, FC head, decoder
    attrs).  Per-op arithmetic is restated and cross-checked against torch CPU.
  * the two TF CTC decoder kernels (greedy, beam search): not in the graphs'
    node lists as code, only as ops with attrs (pinned: beam_width,
    merge_repeated=False, top_paths=1).  PARITY UNPINNED against TF's kernels;
    validated by the reference's own mapping() goldens (greedy) and exhaustive
    CTC enumeration (beam).
"""
