"""TEST INFRASTRUCTURE: CPU restatement of the reference's record-scan path (see exon_oracle.c).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package."""
