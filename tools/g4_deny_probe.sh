for d in 0 1 2 4 8 16 32 64 128; do
  r=$(SMI_GEMM=5ph SMI_G4_DENY=$d python3 tools/fullsize_digest.py --config sd14_512_b1_r4_c3lier --out /tmp/d.pt 2>&1 | grep "digest written" | sed 's/.*\.pt//')
  echo "deny=$d: $r"
done
