"""CPU oracle for the hot path - TEST INFRASTRUCTURE ONLY (see oracle/ops.py header).

The product package ``scratchpad_amd`` never imports this; ``tests/test_no_oracle_in_product.py``
enforces that."""
