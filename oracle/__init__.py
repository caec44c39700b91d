"""CPU oracle of the DeepTables layers hot path — TEST INFRASTRUCTURE ONLY (see reference_layers.py)."""
