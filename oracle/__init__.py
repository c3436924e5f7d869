"""CPU oracle for the path-tracing hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; nothing under slr_amd/ does.
"""
