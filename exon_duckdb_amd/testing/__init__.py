"""Test / bench scaffolding on the Python side (the native part is csrc/testing/ -> libexon_tf_test.so); not the product."""
