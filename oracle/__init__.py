"""CPU restatements of the reference algorithms: TEST INFRASTRUCTURE, never imported by the product."""
