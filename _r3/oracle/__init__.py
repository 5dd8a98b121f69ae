"""
oracle/ -- CPU restatement of the reference (jweyn/DLWP) hot path.  TEST INFRASTRUCTURE ONLY.

Nothing in the product package (dlwp_amd/) may import, call, link or execute anything in here.  The only legitimate
users are tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- and there only as the checker / the
reported CPU baseline, never as the thing measured or shipped.

Pinning status
--------------
* Index / bookkeeping half of the path (PeriodicPadding2D/3D, FillPadding2D, both predict_timeseries variants,
  DataGenerator, delete_nan_samples, the custom loss formulas): PINNED.  np_ref.py is checked bit-for-bit against
  tests/golden/*.npz, which oracle/make_golden.py produced by executing the reference's own source in the build
  container under a numpy stub of its missing third-party imports.
* Conv2D / ZeroPadding2D / MaxPooling2D / UpSampling2D / 'mse' / 'mae' / Adam arithmetic: PARITY UNPINNED.  The
  reference delegates these to standalone Keras 2.2.x + TensorFlow 1.x (unversioned, absent from /root/reference and
  from this image) and holds no tests or golden vectors for them.  np_ref.py restates the Keras-documented semantics
   in float64; torch_ref.py restates the same graph with torch-CPU float32 ops and the two are
  cross-checked against each other in tests/.
"""
