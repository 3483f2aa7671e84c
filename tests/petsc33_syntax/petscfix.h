/* test-only: empty stand-in for the configure-generated petscfix.h (see petscconf.h beside it) */
