"""CPU oracle of the Frustum-PointNet hot path.  TEST INFRASTRUCTURE ONLY: may be
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never
by the fpn package."""
