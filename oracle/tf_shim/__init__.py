"""TEST INFRASTRUCTURE: runs the reference's own Python source on a minimal stand-in for the TensorFlow-1.8 API
(core.py / tfapi.py) -- see core.py.  Used only to pin the oracle; never imported by the product."""
