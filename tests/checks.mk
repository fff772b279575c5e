# tests/checks.mk -- test-only device programs built with the PRODUCT Makefile's compiler, architecture and flags (HIPCC, ARCH, HIPFLAGS, REQUIRED_FLAGS, HDRS
# come from grtrans_amd/csrc/Makefile, which is read first):
#   make -C grtrans_amd/csrc -f Makefile -f ../../tests/checks.mk ../../tests/sqrt_square_check
# (kept out of the product Makefile: that file is part of the device-code identity bench.py checks the committed counter summaries against)
../../tests/sqrt_square_check: ../../tests/sqrt_square_check.hip $(HDRS)
	$(HIPCC) $(HIPFLAGS) $(REQUIRED_FLAGS) ../../tests/sqrt_square_check.hip -o $@
