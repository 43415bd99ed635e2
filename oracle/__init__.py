"""CPU oracle for the dino_predict hot path -- TEST INFRASTRUCTURE ONLY (see oracle/README.md)."""
