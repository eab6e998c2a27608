"""Support code of bench.py (repo root): workloads, roofline objects, BA legs.  Measurement infrastructure, not product."""
