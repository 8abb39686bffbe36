"""CPU oracle for the MMVID video-token hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE.  It restates, in plain fp32 torch-CPU / numpy / C,
the algorithm of the reference (snap-research/MMVID) for the path named in
BASELINE.json.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it, and only as the checker.  Nothing under
``mmvid_amd/`` imports it; the product path fails loudly without its HIP library.

Parity status: PINNED against outputs of the reference itself, run in the build
container by ``tools/make_golden.py`` (fixtures in ``tests/golden/``).  Two pieces of
arithmetic on the path come from third-party packages that are absent from
/root/reference and unpinned in its requirements.txt; for those the oracle follows the
published upstream semantics and is marked "parity unpinned":
  * axial_positional_embedding.AxialPositionalEmbedding (summed mode)
  * torchvision.transforms.RandomErasing (box sampling only; goldens inject the mask)
"""
