"""CPU oracle for the fqtk demux barcode matcher -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product (fqtk_amd) never does.
"""
