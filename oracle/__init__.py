"""CPU oracle for the DAPS measurement-consistency hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it, and only as the checker / the timed CPU baseline.  The product
path (``step_amd``) never imports this package and fails loudly when the HIP
library is missing.

Every function is a PyTorch-CPU restatement of the reference's formulation
(op for op, same dtype promotions) and cites the reference file:line it
follows.  Pinning status (see DESIGN.md "Oracle"):

* pinned against outputs of the reference's own code, imported in the build
  container with third-party stubs (``oracle/gen_golden.py`` ->
  ``tests/golden/*.npz``): everything downstream of ``A_vis`` in
  ``forward_operator/blackhole.py``, the closure index mapping
  (``blackhole.py:235-264``), ``forward_operator/dynamic_mri.py``,
  ``forward_operator/base.py``, ``forward_operator/latent_wrapper.py`` and
  ``sampler/daps.py`` (MCMCSampler);
* PARITY UNPINNED: ``eht_synth`` (uv synthesis, ``ftmatrix`` values, thermal
  sigma, minimal triangle/quadrangle selection) restates ``ehtim==1.2.10``
  from memory; ehtim is neither vendored in the reference nor installed here.
"""
