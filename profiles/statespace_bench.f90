! CPU-only timing of the host state-space routines on the Goutsias model
! (6 species, 10 reactions): rounds of SSA_EXTENDER + ONESTEP_EXTENDER + DROP_STATES as the
! adaptive loop issues them, with a checksum of the resulting FSP.
!   B=krylovfspssa_amd/fortran/_build
!   flang -O3 -fopenmp -I$B profiles/statespace_bench.f90 $B/libkfsp_fortran.a -o /tmp/ssb
!   KFSP_HOST_THREADS=<t> /tmp/ssb <path duration> <rounds> [<dump file>]
! With a dump file the final generator is written as
!   int32 ns, nr, n ; int32 ADJ(nr,n) ; f64 OFFDIAG(nr,n) ; f64 DIAG(n) ; int32 STATE(ns,n)
! for profiles/fsp_spmv_timing.py (SpMV rate on an FSP in the reference's own
! state order).
MODULE SSB_MODEL
  IMPLICIT NONE
CONTAINS
  DOUBLE PRECISION FUNCTION GPROP(STATE, REACTION, PARAMETERS)
    INTEGER, INTENT(IN) :: STATE(:), REACTION
    DOUBLE PRECISION, INTENT(IN), OPTIONAL :: PARAMETERS(:)
    ! species order M, D, RNA, DNA, DNA.D, DNA.2D
    SELECT CASE (REACTION)
    CASE (1);  GPROP = PARAMETERS(1) * STATE(3)
    CASE (2);  GPROP = PARAMETERS(2) * STATE(1)
    CASE (3);  GPROP = PARAMETERS(3) * STATE(5)
    CASE (4);  GPROP = PARAMETERS(4) * STATE(3)
    CASE (5);  GPROP = PARAMETERS(5) * STATE(4) * STATE(2)
    CASE (6);  GPROP = PARAMETERS(6) * STATE(5)
    CASE (7);  GPROP = PARAMETERS(7) * STATE(5) * STATE(2)
    CASE (8);  GPROP = PARAMETERS(8) * STATE(6)
    CASE (9);  GPROP = PARAMETERS(9) * (STATE(1) * (STATE(1) - 1) / 2)
    CASE DEFAULT; GPROP = PARAMETERS(10) * STATE(2)
    END SELECT
  END FUNCTION GPROP
END MODULE SSB_MODEL

PROGRAM SSB
  USE STATESPACE
  USE SSB_MODEL
  IMPLICIT NONE
  TYPE(CME_MODEL) :: MODEL
  TYPE(FINITE_STATE_PROJECTION) :: FSP
  DOUBLE PRECISION :: PAR(10), DT, T0, T1, T2, T3, TS, TO, TD, DSUM
  DOUBLE PRECISION, ALLOCATABLE :: W(:), AW(:)
  LOGICAL :: CHANGED
  INTEGER :: R, NR, NU(6, 10)
  INTEGER(8) :: C0, C1, CR, CHK
  INTEGER :: I, K
  CHARACTER(32) :: ARG
  CHARACTER(256) :: DUMPFILE
  INTEGER :: U
  PAR = (/0.043D0, 0.0007D0, 0.0715D0, 0.0039D0, 0.0199264663575241D0, 0.4791D0, &
          0.000199264663575241D0, 0.8765D-11, 0.0830269431563506104D0, 0.5D0/)
  NU = 0
  NU(1,1) = 1; NU(1,2) = -1; NU(3,3) = 1; NU(3,4) = -1
  NU(4,5) = -1; NU(2,5) = -1; NU(5,5) = 1
  NU(4,6) = 1; NU(2,6) = 1; NU(5,6) = -1
  NU(5,7) = -1; NU(2,7) = -1; NU(6,7) = 1
  NU(5,8) = 1; NU(2,8) = 1; NU(6,8) = -1
  NU(1,9) = -2; NU(2,9) = 1; NU(1,10) = 2; NU(2,10) = -1
  DT = 2.0D0; NR = 40
  IF (COMMAND_ARGUMENT_COUNT() >= 1) THEN
     CALL GET_COMMAND_ARGUMENT(1, ARG); READ(ARG, *) DT
  ENDIF
  IF (COMMAND_ARGUMENT_COUNT() >= 2) THEN
     CALL GET_COMMAND_ARGUMENT(2, ARG); READ(ARG, *) NR
  ENDIF
  DUMPFILE = ''
  IF (COMMAND_ARGUMENT_COUNT() >= 3) CALL GET_COMMAND_ARGUMENT(3, DUMPFILE)
  CALL MODEL%CREATE(6, 10, 10)
  MODEL%STOICHIOMETRY = NU
  MODEL%CUSTOMPROP => GPROP
  CALL MODEL%RESET_PARAMETERS(PAR)
  MODEL%LOADED = .TRUE.
  CALL FSP%CREATE(MODEL)
  FSP%SIZE = 1
  FSP%STATE(:, 1) = [2, 6, 0, 2, 0, 0]
  CALL MATRIX_STARTER(FSP, MODEL)
  CALL SYSTEM_CLOCK(COUNT_RATE=CR)
  TS = 0; TO = 0; TD = 0
  DO R = 1, NR
     CALL SYSTEM_CLOCK(C0)
     CALL SSA_EXTENDER(DT, FSP, MODEL)
     CALL SYSTEM_CLOCK(C1)
     T1 = DBLE(C1 - C0) / CR
     I = FSP%SIZE
     CALL ONESTEP_EXTENDER(FSP, MODEL)
     CALL SYSTEM_CLOCK(C0)
     T2 = DBLE(C0 - C1) / CR
     TS = TS + T1; TO = TO + T2
     ! a quarter of the states falls below the threshold and is dropped, as after
     ! an accepted step of the solver (the two extenders bring the border back)
     T3 = 0
     K = FSP%SIZE
     IF (K > 20000) THEN
        ALLOCATE(W(K), AW(K))
        AW = 0.0D0
        W = 1.0D0 / K
        W(1:K:4) = 1.0D-14
        DSUM = 1.0D0
        CALL SYSTEM_CLOCK(C0)
        CALL DROP_STATES_CORE(W, FSP, MODEL, DSUM, AW, CHANGED)
        CALL SYSTEM_CLOCK(C1)
        T3 = DBLE(C1 - C0) / CR
        TD = TD + T3
        DEALLOCATE(W, AW)
     ENDIF
     WRITE(*, '(I4,3I10,3F9.3)') R, I, K, FSP%SIZE, T1, T2, T3
     IF (FSP%SIZE > 3000000) EXIT
  ENDDO
  CHK = 0
  DO I = 1, FSP%SIZE
     DO K = 1, 10
        CHK = CHK * 31_8 + FSP%MATRIX%ADJ(K, I)
     ENDDO
     CHK = CHK * 17_8 + FSP%STATE(1, I) + 7 * FSP%STATE(2, I) + 13 * FSP%STATE(3, I)
  ENDDO
  WRITE(*, '(A,F9.3,A,F9.3,A,F9.3,A,I10,A,I22)') 'ssa ', TS, ' onestep ', TO, ' drop ', TD, ' n ', FSP%SIZE, ' chk ', CHK
  WRITE(*, '(A,9F8.3)') 'passes: onestep scan/append/link, ssa walk/link, drop flags/compact/renumber/table', STATESPACE_SEC
  IF (LEN_TRIM(DUMPFILE) > 0) THEN
     OPEN(NEWUNIT=U, FILE=TRIM(DUMPFILE), ACCESS='STREAM', FORM='UNFORMATTED', STATUS='REPLACE')
     WRITE(U) 6, 10, FSP%SIZE
     WRITE(U) FSP%MATRIX%ADJ(1:10, 1:FSP%SIZE)
     WRITE(U) FSP%MATRIX%OFFDIAG(1:10, 1:FSP%SIZE)
     WRITE(U) FSP%MATRIX%DIAG(1:FSP%SIZE)
     WRITE(U) FSP%STATE(1:6, 1:FSP%SIZE)
     CLOSE(U)
  ENDIF
END PROGRAM SSB
