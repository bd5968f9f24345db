"""
oracle/ -- CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.

This package is the *checker* for the HIP path, never the thing shipped or
measured: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it.  ``sdfs_via_autodiff_amd`` never imports
anything from here and fails loudly when its HIP library is missing.

What is restated (reference paths are relative to /root/reference):

* ``rouwenhorst.py``  quantecon.markov.approximation.rouwenhorst (third-party,
  not vendored by the reference, version unpinned -- no requirements file);
  call sites code/ssy/discrete/ssy_wc_ratio.py:48-50,63 and
  code/gcy/discrete/gcy_wc_ratio.py:65-68,97,115.
* ``models.py``       SSY / GCY default calibrations and ``params`` tuple order
  (code/ssy/ssy_model.py:57-81, code/gcy/gcy_model.py:45-75).
* ``ssy.py``          discretize_ssy, T_ssy (literal O(N^2) sum), T_ssy_loops,
  factorised T and analytic JVP (code/ssy/discrete/ssy_wc_ratio.py:23-199).
* ``gcy.py``          discretize_gcy, T_gcy, T_gcy_loops, factorised T and JVP
  (code/gcy/discrete/gcy_wc_ratio.py:31-302).
* ``solvers.py``      successive_approx / newton_solver / anderson_solver /
  solver front end (code/solvers.py:19-177).
* ``c/``              plain C + OpenMP factorised operator (same arithmetic as
  the numpy factorised path) used as the multi-core CPU baseline.

Parity pinning (see tests/test_oracle_golden.py, tests/golden/make_golden.py):
the restatement is checked against golden vectors produced by importing the
reference's own discrete modules in the build container (numpy standing in for
jax.numpy; script committed) and against the one recorded output the reference
holds (code/ssy/discrete/sandpit.ipynb:41-44).  jaxopt's Anderson iterate
sequence has no recorded output anywhere in the reference: its iterate-level
parity is UNPINNED; only its fixed point is checked.
"""
