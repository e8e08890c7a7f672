! TEST INFRASTRUCTURE - NOT PRODUCT CODE.
!
! ref_dump: a driver of OUR OWN that links against the *unmodified* reference
! modules compiled from /root/reference by oracle/Makefile (objects live only in
! oracle/_ref/, git-ignored) and writes what the reference computes to small
! binary files.  oracle/make_golden.py turns those into tests/golden/*.npz.
!
! It only touches public entities of the reference:
!   MODELMODULE  : CME_MODEL%LOAD/CREATE/RESET_PARAMETERS/PROPENSITY, CUSTOMPROP
!                  (src/model/ModelModule.f90:14-42)
!   STATESPACE   : FINITE_STATE_PROJECTION%CREATE, MATRIX_STARTER,
!                  ONESTEP_EXTENDER, SSA_EXTENDER, FIND_DROPTOL, DROP_STATES
!                  (src/state_space/StateSpace.f90:19-45,248,347,398,431,550)
!   KRYLOVSOLVER : CME_SOLVE (src/fsp/KrylovSolver.f90:7)
!   DGPADM       : src/expokit/dgpadm.f:2
!
! usage:  ref_dump assembly <toggle|repressilator|goutsias> <k> <out.bin>
!         ref_dump ssa      <toggle|repressilator|goutsias> <k> <dt> <out.bin>
!         ref_dump drop     <toggle|repressilator|goutsias> <k> <dsum> <out.bin>
!         ref_dump droptol  <out.bin>
!         ref_dump solve    <case> <out.bin> [T]    (log goes to stdout)
!         ref_dump padm     <out.bin>
!         ref_dump proptable <out.bin>
!         ref_dump exprtable <out.bin>
!         ref_dump api      <out.bin>
!
! binary layout (stream, native endian):
!   fsp file : int32 ns, nr, n ; int32 STATE(ns,n) ; int32 ADJ(nr,n) ;
!              f64 OFFDIAG(nr,n) ; f64 DIAG(n) ; f64 VECTOR(n)
MODULE REF_DUMP_MATVEC
  ! y = A x in the scatter form of the column layout (what the solver hands to
  ! DROP_STATES as its FMATVEC argument, KrylovSolver.f90:511,577-607)
  USE STATESPACE
  IMPLICIT NONE
CONTAINS
  SUBROUTINE DUMP_MATVEC(X, Y, MATRIX)
    DOUBLE PRECISION :: X(*), Y(*)
    TYPE(FSP_MATRIX) :: MATRIX
    INTEGER :: I, K, J
    DO I = 1, MATRIX%SIZE
       Y(I) = 0.0D0
    ENDDO
    DO I = 1, MATRIX%SIZE
       Y(I) = Y(I) - MATRIX%DIAG(I) * X(I)
       DO K = 1, SIZE(MATRIX%ADJ, 1)
          J = MATRIX%ADJ(K, I)
          IF (J > 0) Y(J) = Y(J) + MATRIX%OFFDIAG(K, I) * X(I)
       ENDDO
    ENDDO
  END SUBROUTINE DUMP_MATVEC
END MODULE REF_DUMP_MATVEC

PROGRAM REF_DUMP
  USE REF_CASES
  USE REF_DUMP_MATVEC
  USE STATESPACE
  USE KRYLOVSOLVER
  IMPLICIT NONE

  CHARACTER(LEN=256) :: MODE, ARG2, ARG3, ARG4, ARG5

  CALL GET_COMMAND_ARGUMENT(1, MODE)
  CALL GET_COMMAND_ARGUMENT(2, ARG2)
  CALL GET_COMMAND_ARGUMENT(3, ARG3)
  CALL GET_COMMAND_ARGUMENT(4, ARG4)
  CALL GET_COMMAND_ARGUMENT(5, ARG5)

  SELECT CASE (TRIM(MODE))
  CASE ('assembly')
     CALL DO_ASSEMBLY(TRIM(ARG2), TRIM(ARG3), TRIM(ARG4))
  CASE ('ssa')
     CALL DO_SSA(TRIM(ARG2), TRIM(ARG3), TRIM(ARG4), TRIM(ARG5))
  CASE ('drop')
     CALL DO_DROP(TRIM(ARG2), TRIM(ARG3), TRIM(ARG4), TRIM(ARG5))
  CASE ('droptol')
     CALL DO_DROPTOL(TRIM(ARG2))
  CASE ('solve')
     CALL DO_SOLVE(TRIM(ARG2), TRIM(ARG3))
  CASE ('padm')
     CALL DO_PADM(TRIM(ARG2))
  CASE ('proptable')
     CALL DO_PROPTABLE(TRIM(ARG2))
  CASE ('exprtable')
     CALL DO_EXPRTABLE(TRIM(ARG2))
  CASE ('api')
     CALL DO_API(TRIM(ARG2))
  CASE DEFAULT
     STOP 'ref_dump: unknown mode'
  END SELECT

CONTAINS

  !---------------------------------------------------------------- G1
  SUBROUTINE DO_ASSEMBLY(NAME, KSTR, FNAME)
    CHARACTER(LEN=*), INTENT(IN) :: NAME, KSTR, FNAME
    TYPE(CME_MODEL) :: MODEL
    TYPE(FINITE_STATE_PROJECTION) :: FSP
    INTEGER, ALLOCATABLE :: X0(:)
    INTEGER :: K, I
    READ(KSTR, *) K
    CALL LOAD_INPUT_MODEL(NAME, MODEL, X0)
    CALL FSP%CREATE(MODEL, TABLEN)
    FSP%SIZE = 1
    FSP%STATE(:, 1) = X0
    FSP%VECTOR = 0.0D0
    FSP%VECTOR(1) = 1.0D0
    CALL MATRIX_STARTER(FSP, MODEL)
    DO I = 1, K
       CALL ONESTEP_EXTENDER(FSP, MODEL)
    ENDDO
    CALL WRITE_FSP(FNAME, MODEL, FSP)
    PRINT *, 'ASSEMBLY ', NAME, ' K=', K, ' N=', FSP%SIZE
  END SUBROUTINE DO_ASSEMBLY

  !---------------------------------------------------------------- G1b
  ! k rounds of SSA_EXTENDER(dt) + ONESTEP_EXTENDER, as the solver issues them
  ! (KrylovSolver.f90:528-529), on the default random stream.  One more uniform
  ! number is drawn at the end and stored in VECTOR(1): it pins how many numbers
  ! the paths consumed.
  SUBROUTINE DO_SSA(NAME, KSTR, DTSTR, FNAME)
    CHARACTER(LEN=*), INTENT(IN) :: NAME, KSTR, DTSTR, FNAME
    TYPE(CME_MODEL) :: MODEL
    TYPE(FINITE_STATE_PROJECTION) :: FSP
    INTEGER, ALLOCATABLE :: X0(:)
    INTEGER :: K, I
    DOUBLE PRECISION :: DT, R
    READ(KSTR, *) K
    READ(DTSTR, *) DT
    CALL LOAD_INPUT_MODEL(NAME, MODEL, X0)
    CALL FSP%CREATE(MODEL, TABLEN)
    FSP%SIZE = 1
    FSP%STATE(:, 1) = X0
    FSP%VECTOR = 0.0D0
    CALL MATRIX_STARTER(FSP, MODEL)
    DO I = 1, K
       CALL SSA_EXTENDER(DT, FSP, MODEL)
       CALL ONESTEP_EXTENDER(FSP, MODEL)
    ENDDO
    CALL RANDOM_NUMBER(R)
    FSP%VECTOR(1) = R
    CALL WRITE_FSP(FNAME, MODEL, FSP)
    PRINT *, 'SSA ', NAME, ' K=', K, ' DT=', DT, ' N=', FSP%SIZE
  END SUBROUTINE DO_SSA

  !---------------------------------------------------------------- G1c
  ! a deterministic positive vector spanning 1e-2 .. 1e-16 (golden-ratio sequence)
  SUBROUTINE SPREAD_VECTOR(N, W)
    INTEGER, INTENT(IN) :: N
    DOUBLE PRECISION, INTENT(OUT) :: W(:)
    INTEGER :: I
    DOUBLE PRECISION :: U
    DO I = 1, N
       U = I * 0.6180339887498949D0
       U = U - AINT(U)
       W(I) = 10.0D0**(-2.0D0 - 14.0D0 * U)
    ENDDO
  END SUBROUTINE SPREAD_VECTOR

  ! DROP_STATES (StateSpace.f90:431-548) on the FSP of the assembly case with a
  ! decaying vector, then one ONESTEP_EXTENDER on what is left; the vector
  ! after the drop is dumped in VECTOR
  SUBROUTINE DO_DROP(NAME, KSTR, DSTR, FNAME)
    CHARACTER(LEN=*), INTENT(IN) :: NAME, KSTR, DSTR, FNAME
    TYPE(CME_MODEL) :: MODEL
    TYPE(FINITE_STATE_PROJECTION) :: FSP
    INTEGER, ALLOCATABLE :: X0(:)
    DOUBLE PRECISION, ALLOCATABLE :: W(:)
    INTEGER :: K, I, N1
    DOUBLE PRECISION :: DSUM
    READ(KSTR, *) K
    READ(DSTR, *) DSUM
    CALL LOAD_INPUT_MODEL(NAME, MODEL, X0)
    CALL FSP%CREATE(MODEL, TABLEN)
    FSP%SIZE = 1
    FSP%STATE(:, 1) = X0
    FSP%VECTOR = 0.0D0
    CALL MATRIX_STARTER(FSP, MODEL)
    DO I = 1, K
       CALL ONESTEP_EXTENDER(FSP, MODEL)
    ENDDO
    ALLOCATE(W(TABLEN))
    W = 0.0D0
    ! decays over 18 decades along the list (breadth-first order: neighbours
    ! have similar magnitudes, as in a solve), with every 7th entry raised so
    ! that kept and dropped states interleave
    N1 = FSP%SIZE
    DO I = 1, N1
       W(I) = 10.0D0**(-2.0D0 - 18.0D0 * DBLE(I - 1) / DBLE(MAX(N1 - 1, 1)))
       IF (MOD(I, 7) == 0) W(I) = W(I) * 1.0D3
    ENDDO
    PRINT *, 'N BEFORE DROP=', N1
    CALL DROP_STATES(W, FSP, MODEL, DSUM, DUMP_MATVEC)
    N1 = FSP%SIZE
    FSP%VECTOR(1:N1) = W(1:N1)
    CALL ONESTEP_EXTENDER(FSP, MODEL)
    CALL WRITE_FSP(FNAME, MODEL, FSP)
    PRINT *, 'DROP ', NAME, ' K=', K, ' DSUM=', DSUM, ' N AFTER DROP=', N1, ' N=', FSP%SIZE
  END SUBROUTINE DO_DROP

  ! FIND_DROPTOL (StateSpace.f90:398-427) for a range of mass bounds
  SUBROUTINE DO_DROPTOL(FNAME)
    CHARACTER(LEN=*), INTENT(IN) :: FNAME
    INTEGER, PARAMETER :: N = 50000, ND = 10
    DOUBLE PRECISION :: W(N), TOL(ND), DS(ND)
    INTEGER :: I, U
    CALL SPREAD_VECTOR(N, W)
    W(7:N:13) = 0.0D0
    W(5:N:17) = -W(5:N:17)
    W(3:N:11) = W(3:N:11) * 1.0D-12
    DS = [1.0D-1, 1.0D-4, 1.0D-6, 1.0D-8, 1.0D-10, 1.0D-13, 1.0D-16, 1.0D-20, 1.0D-24, 1.0D-27]
    DO I = 1, ND
       CALL FIND_DROPTOL(6, N, W, TOL(I), DS(I))
    ENDDO
    OPEN(NEWUNIT=U, FILE=FNAME, ACCESS='STREAM', FORM='UNFORMATTED', STATUS='REPLACE')
    WRITE(U) DS, TOL
    CLOSE(U)
    PRINT *, 'DROPTOL ', TOL
  END SUBROUTINE DO_DROPTOL

  !---------------------------------------------------------------- G3 / G6
  SUBROUTINE DO_SOLVE(CASENAME, FNAME)
    CHARACTER(LEN=*), INTENT(IN) :: CASENAME, FNAME
    TYPE(CME_MODEL) :: MODEL
    TYPE(FINITE_STATE_PROJECTION) :: FSP_IN, FSP
    DOUBLE PRECISION :: T, FSPTOL, KRYTOL
    INTEGER :: N
    ! the reference's drivers call this first (TestSolverFromFile.f90:19)
    CALL RANDOM_SEED()

    CALL SETUP_SOLVE_CASE(CASENAME, MODEL, FSP_IN, FSP, T, FSPTOL, KRYTOL)

    ! optional 4th argument overrides the end time (used when bisecting a
    ! trajectory against the restatement)
    IF (LEN_TRIM(ARG4) > 0) READ(ARG4, *) T

    ! inputs first (so the fixture holds p0 in the seeded order)
    CALL WRITE_FSP(FNAME // '.in', MODEL, FSP_IN)
    CALL CME_SOLVE(MODEL, T, FSP_IN, FSP, FSPTOL, KRYTOL, VERBOSITY = 1)
    CALL WRITE_FSP(FNAME, MODEL, FSP)
    N = FSP%SIZE
    PRINT *, 'FINAL SIZE', N
    PRINT *, 'FINAL SUM ', SUM(FSP%VECTOR(1:N))
  END SUBROUTINE DO_SOLVE

  !---------------------------------------------------------------- G5
  SUBROUTINE DO_PADM(FNAME)
    ! exp(t*H) by the reference DGPADM (dgpadm.f:2-169, IDEG = 6 as
    ! KrylovSolver.f90:82) for a few deterministic Hessenberg-like matrices.
    ! file: int32 ncase ; per case: int32 m ; f64 t ; f64 H(m,m) ; int32 ns ;
    !       f64 E(m,m)
    CHARACTER(LEN=*), INTENT(IN) :: FNAME
    INTEGER, PARAMETER :: NCASE = 5
    INTEGER :: MS(NCASE), U, C, M, I, J, IEXPH, NS, IFLAG
    DOUBLE PRECISION :: TS(NCASE)
    DOUBLE PRECISION, ALLOCATABLE :: H(:, :), WSP(:)
    INTEGER, ALLOCATABLE :: IPIV(:)
    MS = [3, 12, 32, 62, 102]
    TS = [1.0D0, 0.05D0, 0.5D0, 2.0D0, 0.3D0]
    OPEN(NEWUNIT=U, FILE=FNAME, ACCESS='STREAM', FORM='UNFORMATTED', STATUS='REPLACE')
    WRITE(U) NCASE
    DO C = 1, NCASE
       M = MS(C)
       ALLOCATE(H(M, M), WSP(4 * M * M + 7), IPIV(M))
       H = 0.0D0
       ! banded upper part + one subdiagonal, like the IOP Hessenberg
       ! (KrylovSolver.f90:241-257), entries from a fixed recurrence
       DO J = 1, M
          DO I = MAX(1, J - 1), MIN(M, J + 1)
             H(I, J) = 3.0D0 * SIN(1.7D0 * I + 0.3D0 * J * C) - 2.0D0 * (I - J) &
                  - MERGE(8.0D0, 0.0D0, I == J)
          ENDDO
       ENDDO
       IF (M > 2) H(M, M - 1) = 1.0D0      ! the corrected-scheme corner (:266)
       CALL DGPADM(6, M, TS(C), H, M, WSP, 4 * M * M + 7, IPIV, IEXPH, NS, IFLAG)
       WRITE(U) M
       WRITE(U) TS(C)
       WRITE(U) H
       WRITE(U) NS
       WRITE(U) WSP(IEXPH:IEXPH + M * M - 1)
       DEALLOCATE(H, WSP, IPIV)
    ENDDO
    CLOSE(U)
  END SUBROUTINE DO_PADM

  !---------------------------------------------------------------- G4
  SUBROUTINE DO_PROPTABLE(FNAME)
    ! the 50x50x4 propensity table of test/TestModelParser.f90:33-43, evaluated
    ! by the reference's bytecode parser on models/toggle_test_model.input.
    ! file: f64 P(4,50,50)  (reaction fastest, then j, then i)
    CHARACTER(LEN=*), INTENT(IN) :: FNAME
    TYPE(CME_MODEL) :: MODEL
    DOUBLE PRECISION :: P(4, 50, 50)
    INTEGER :: I, J, R, U
    CALL MODEL%LOAD('toggle_test_model.input')
    CALL MODEL%RESET_PARAMETERS((/5000.0D0, 1600.D0, 1.0D0, 1.0D0/))
    DO I = 1, 50
       DO J = 1, 50
          DO R = 1, 4
             P(R, J, I) = MODEL%PROPENSITY((/I, J/), R)
          ENDDO
       ENDDO
    ENDDO
    OPEN(NEWUNIT=U, FILE=FNAME, ACCESS='STREAM', FORM='UNFORMATTED', STATUS='REPLACE')
    WRITE(U) P
    CLOSE(U)
  END SUBROUTINE DO_PROPTABLE

  SUBROUTINE DO_EXPRTABLE(FNAME)
    ! propensities of expr_test_model.input (a model file of OURS: operator
    ! precedence and associativity, unary minus, '**', every function of the
    ! expression type, D/E exponents, a species name with '.' and digits,
    ! division by zero and log of a non-positive number) on a 13 x 13 x 3 grid.
    ! file: f64 P(16,3,13,13)  (reaction fastest, then DNA.2D, Y, X)
    CHARACTER(LEN=*), INTENT(IN) :: FNAME
    TYPE(CME_MODEL) :: MODEL
    DOUBLE PRECISION :: P(16, 3, 13, 13)
    INTEGER :: I, J, K, R, U
    CALL MODEL%LOAD('expr_test_model.input')
    CALL MODEL%RESET_PARAMETERS((/7.5D0, 2.0D0, 0.75D0, 0.3D0, 4.0D0/))
    DO I = 0, 12
       DO J = 0, 12
          DO K = 0, 2
             DO R = 1, 16
                P(R, K + 1, J + 1, I + 1) = MODEL%PROPENSITY((/I, J, K/), R)
             ENDDO
          ENDDO
       ENDDO
    ENDDO
    OPEN(NEWUNIT=U, FILE=FNAME, ACCESS='STREAM', FORM='UNFORMATTED', STATUS='REPLACE')
    WRITE(U) P
    CLOSE(U)
    ! the stoichiometry LOAD read from the reaction strings ('2X -> Y', 'X -> DNA.2D', ...)
    OPEN(NEWUNIT=U, FILE=FNAME // '.stoich', ACCESS='STREAM', FORM='UNFORMATTED', STATUS='REPLACE')
    WRITE(U) MODEL%NSPECIES, MODEL%NREACTIONS
    WRITE(U) MODEL%STOICHIOMETRY(1:MODEL%NSPECIES, 1:MODEL%NREACTIONS)
    CLOSE(U)
    PRINT *, 'EXPRTABLE ', MODEL%NSPECIES, MODEL%NREACTIONS, P(:, 1, 5, 8)
  END SUBROUTINE DO_EXPRTABLE

  SUBROUTINE DO_API(FNAME)
    ! the bound procedures of FINITE_STATE_PROJECTION a driver may call itself
    ! (StateSpace.f90:19-45): ADD (repeated, unordered, one state twice, a negative one),
    ! INDEX and PROBABILITY of listed and unlisted states.  The FSP goes to FNAME, the
    ! query results to FNAME.q (int32 idx(nq), f64 prob(nq)).
    CHARACTER(LEN=*), INTENT(IN) :: FNAME
    TYPE(CME_MODEL) :: MODEL
    TYPE(FINITE_STATE_PROJECTION) :: FSP
    INTEGER, ALLOCATABLE :: X0(:)
    INTEGER, PARAMETER :: NQ = 8
    INTEGER :: I, U, Q(2, NQ), IDX(NQ), ST(2)
    DOUBLE PRECISION :: PR(NQ)
    CALL LOAD_INPUT_MODEL('toggle', MODEL, X0)
    CALL FSP%CREATE(MODEL, TABLEN)
    FSP%SIZE = 1
    FSP%STATE(:, 1) = X0
    FSP%VECTOR = 0.0D0
    CALL MATRIX_STARTER(FSP, MODEL)
    DO I = 1, 3
       CALL ONESTEP_EXTENDER(FSP, MODEL)
    ENDDO
    ST = [7, 2];  CALL FSP%ADD(MODEL, ST)
    ST = [1, 4];  CALL FSP%ADD(MODEL, ST)
    ST = [7, 2];  CALL FSP%ADD(MODEL, ST)          ! already listed
    ST = [6, 2];  CALL FSP%ADD(MODEL, ST)          ! neighbour of a listed one
    ST = [-1, 3]; CALL FSP%ADD(MODEL, ST)          ! not a state
    ST = [0, 4];  CALL FSP%ADD(MODEL, ST)
    DO I = 1, FSP%SIZE
       FSP%VECTOR(I) = 1.0D0 / DBLE(I + 1)
    ENDDO
    Q = RESHAPE([0, 0, 7, 2, 1, 4, 6, 2, 5, 5, -1, 3, 0, 4, 2, 1], [2, NQ])
    DO I = 1, NQ
       IDX(I) = FSP%INDEX(Q(:, I))
       PR(I) = FSP%PROBABILITY(Q(:, I))
    ENDDO
    CALL WRITE_FSP(FNAME, MODEL, FSP)
    OPEN(NEWUNIT=U, FILE=FNAME // '.q', ACCESS='STREAM', FORM='UNFORMATTED', STATUS='REPLACE')
    WRITE(U) IDX
    WRITE(U) PR
    CLOSE(U)
    PRINT *, 'API N=', FSP%SIZE, ' IDX=', IDX
  END SUBROUTINE DO_API

END PROGRAM REF_DUMP
