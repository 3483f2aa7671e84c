/* TEST-ONLY compile aid (tests/test_integration_shim.py::test_petsc_flavour_compiles_against_the_reference_headers).
 *
 * PETSc's public and private headers include a configure-generated petscconf.h; ./configure cannot run here (Python 2,
 * un-vendored BuildSystem).  This file supplies the handful of macros those HEADERS need so that gcc -fsyntax-only can
 * parse the plug-in's OWN sources (petsc-dev_amd/host/*.c, integration/petsc-3.3/*) in their -DPETSCHIPMI355X_WITH_PETSC
 * flavour against /root/reference/include.  Nothing of the reference is compiled or linked with it, no object file is
 * produced, it is not an oracle and pins nothing; it only proves the sources are well-formed C against the real
 * struct _p_Vec / _p_Mat / _VecOps / _MatOps / PetscError / SETERRQ declarations (MPIUNI flavour: include/mpiuni/mpi.h). */
#if !defined(INCLUDED_PETSCCONF_H)
#define INCLUDED_PETSCCONF_H
#define PETSC_ARCH "probe"
#define PETSC_DIR "/root/reference"
#define PETSC_LIB_DIR "unused"
#define PETSC_HAVE_MPIUNI 1
#define PETSC_USE_REAL_DOUBLE 1
#define PETSC_USE_SCALAR_REAL 1
#define PETSC_CLANGUAGE_C 1
#define PETSC_USE_SINGLE_LIBRARY 1
#define PETSC_SIZEOF_INT 4
#define PETSC_SIZEOF_LONG 8
#define PETSC_SIZEOF_LONG_LONG 8
#define PETSC_SIZEOF_VOID_P 8
#define PETSC_SIZEOF_SHORT 2
#define PETSC_SIZEOF_CHAR 1
#define PETSC_SIZEOF_FLOAT 4
#define PETSC_SIZEOF_DOUBLE 8
#define PETSC_SIZEOF_SIZE_T 8
#define PETSC_BITS_PER_BYTE 8
#define PETSC_MEMALIGN 16
#define PETSC_LEVEL1_DCACHE_LINESIZE 64
#define PETSC_UINTPTR_T uintptr_t
#define PETSC_FUNCTION_NAME_C __func__
#define PETSC_FUNCTION_NAME_CXX __func__
#define PETSC_RESTRICT __restrict__
#define PETSC_STATIC_INLINE static inline
#define PETSC_HAVE_BUILTIN_EXPECT 1
#define PETSC_HAVE_STDLIB_H 1
#define PETSC_HAVE_STRING_H 1
#define PETSC_HAVE_STDINT_H 1
#define PETSC_HAVE_UNISTD_H 1
#define PETSC_HAVE_SYS_TIME_H 1
#define PETSC_HAVE_SYS_TYPES_H 1
#define PETSC_HAVE_MALLOC_H 1
#define PETSC_HAVE_MATH_H 1
#define PETSC_HAVE_FLOAT_H 1
#define PETSC_HAVE_LIMITS_H 1
#define PETSC_HAVE_TIME_H 1
#define PETSC_HAVE_STDARG_H 1
#define PETSC_HAVE_GETTIMEOFDAY 1
#define PETSC_HAVE_ISNAN 1
#define PETSC_HAVE_ISINF 1
#define PETSC_HAVE_VA_COPY 1
#define PETSC_HAVE_VSNPRINTF 1
#define PETSC_HAVE_MEMMOVE 1
#define PETSC_HAVE_DOUBLE_ALIGN_MALLOC 1
#define PETSC_USE_GETTIMEOFDAY 1
#define PETSC_BLASLAPACK_UNDERSCORE 1
#define PETSC_HAVE_XMMINTRIN_H 1
#define PETSC_Prefetch(a,b,c) _mm_prefetch((const char*)(a),(c))
#define PETSC_PREFETCH_HINT_NTA _MM_HINT_NTA
#define PETSC_PREFETCH_HINT_T0 _MM_HINT_T0
#define PETSC_PREFETCH_HINT_T1 _MM_HINT_T1
#define PETSC_PREFETCH_HINT_T2 _MM_HINT_T2
#define PETSC_USE_PROC_FOR_SIZE 1
#define PETSC_RETSIGTYPE void
#define PETSC_HAVE_SIGNAL 1
#define PETSC_HAVE_SIGNAL_H 1
#define PETSC_USE_INFO 1
#define PETSC_USE_LOG 1
#define PETSC_USE_CTABLE 1
#define PETSC_USE_BACKWARD_LOOP 1
#define PETSC_HAVE_RAND 1
#define PETSC_HAVE_DRAND48 1
#define PETSC_USE_ERRORCHECKING 1
#define PETSC_IS_COLOR_VALUE_TYPE short
#define PETSC_UNUSED __attribute((unused))
#define IS_COLORING_MAX 65535
#define MPIU_COLORING_VALUE MPI_UNSIGNED_SHORT
#define PETSC_SLSUFFIX "so"
#define PETSC_USE_GDB_DEBUGGER 1
#define PETSC_PATH_SEPARATOR ':'
#define PETSC_DIR_SEPARATOR '/'
#define PETSC_REPLACE_DIR_SEPARATOR '\\'
#endif
