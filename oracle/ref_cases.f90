! TEST INFRASTRUCTURE - NOT PRODUCT CODE.
!
! The workloads the dump / trace / replay drivers share: model set-up with the
! parameter values of the reference's own drivers, the seed FSPs, and the named
! CME_SOLVE cases.  Only public entities of MODELMODULE / STATESPACE are used, so
! this file compiles unchanged against the reference's modules (oracle/Makefile)
! and against ours (krylovfspssa_amd/fortran/Makefile).
MODULE REF_CASES
  USE STATESPACE
  IMPLICIT NONE
  INTEGER, PARAMETER :: TABLEN = 100009   ! prime, see StateSpace.f90:9
  ! closed-ring test model (module procedure: the pointer stored in
  ! CME_MODEL%CUSTOMPROP needs no trampoline)
  DOUBLE PRECISION :: RING_CF(6), RING_CB(6)
  INTEGER :: RING_NS
CONTAINS
  DOUBLE PRECISION FUNCTION RING_PROP(STATE, REACTION, PARAMETERS)
    INTEGER, INTENT(IN) :: STATE(:), REACTION
    DOUBLE PRECISION, INTENT(IN), OPTIONAL :: PARAMETERS(:)
    INTEGER :: I, K
    I = (REACTION + 1) / 2
    K = MOD(I, RING_NS) + 1
    IF (MOD(REACTION, 2) == 1) THEN
       RING_PROP = RING_CF(I) * STATE(I)
    ELSE
       RING_PROP = RING_CB(I) * STATE(K)
    ENDIF
  END FUNCTION RING_PROP

  DOUBLE PRECISION FUNCTION TOGGLE_EXAMPLE_PROP(STATE, REACTION, PARAMETERS)
    ! the propensities of examples/toggle.f90:55-69, restated
    INTEGER, INTENT(IN) :: STATE(:), REACTION
    DOUBLE PRECISION, INTENT(IN), OPTIONAL :: PARAMETERS(:)
    TOGGLE_EXAMPLE_PROP = 0.0D0
    IF (REACTION == 1) TOGGLE_EXAMPLE_PROP = PARAMETERS(1) + PARAMETERS(2) / (1D0 + STATE(2)**1.5D0)
    IF (REACTION == 2) TOGGLE_EXAMPLE_PROP = PARAMETERS(3) * STATE(1)
    IF (REACTION == 3) TOGGLE_EXAMPLE_PROP = PARAMETERS(4) + PARAMETERS(5) / (1D0 + STATE(1)**3.5D0)
    IF (REACTION == 4) TOGGLE_EXAMPLE_PROP = PARAMETERS(6) * STATE(2)
  END FUNCTION TOGGLE_EXAMPLE_PROP

  DOUBLE PRECISION FUNCTION REPRESSILATOR_EXAMPLE_PROP(STATE, REACTION, PARAMETERS)
    ! the propensities of examples/repressilator.f90:50-69, restated (Hill exponent 6, births and deaths interleaved)
    INTEGER, INTENT(IN) :: STATE(:), REACTION
    DOUBLE PRECISION, INTENT(IN), OPTIONAL :: PARAMETERS(:)
    INTEGER :: S, R
    S = (REACTION + 1) / 2                          ! the species reaction 2S-1 makes and 2S removes
    R = MOD(S, 3) + 1                               ! its repressor: 2, 3, 1
    IF (MOD(REACTION, 2) == 1) THEN
       REPRESSILATOR_EXAMPLE_PROP = PARAMETERS(1) / (1D0 + PARAMETERS(2) * STATE(R)**6.0D0)
    ELSE
       REPRESSILATOR_EXAMPLE_PROP = PARAMETERS(3) * STATE(S)
    ENDIF
  END FUNCTION REPRESSILATOR_EXAMPLE_PROP

  DOUBLE PRECISION FUNCTION REPRESSILATOR_VARIANT_PROP(STATE, REACTION, PARAMETERS)
    ! Repressilator-shaped networks of OUR OWN that exercise how a compiled-in function is probed (KFSP_CUSTOMPROP of the
    ! Fortran host; for the reference these are just three more CUSTOMPROP models).  PARAMETERS(4) selects the production law:
    !   1  "pair":   repressed by one species, damped by the other - depends on TWO species, not a product
    !   2  "triple": depends on all three species, not a product   (no device plan: the host keeps the propensities)
    !   3  "trap":   as examples/repressilator.f90 except on the plane X3 = 30, which no probe visits - a plan the probe
    !                accepts and the final verification must reject
    INTEGER, INTENT(IN) :: STATE(:), REACTION
    DOUBLE PRECISION, INTENT(IN), OPTIONAL :: PARAMETERS(:)
    INTEGER :: S, R, Q
    S = (REACTION + 1) / 2
    R = MOD(S, 3) + 1
    Q = MOD(R, 3) + 1
    IF (MOD(REACTION, 2) == 0) THEN
       REPRESSILATOR_VARIANT_PROP = PARAMETERS(3) * STATE(S)
       RETURN
    ENDIF
    SELECT CASE (NINT(PARAMETERS(4)))
    CASE (1)
       REPRESSILATOR_VARIANT_PROP = PARAMETERS(1) / (1D0 + PARAMETERS(2) * STATE(R)**2.0D0 + 0.5D0 * STATE(Q))
    CASE (2)
       REPRESSILATOR_VARIANT_PROP = PARAMETERS(1) / (1D0 + PARAMETERS(2) * STATE(R)**2.0D0 + 0.5D0 * STATE(Q) + 0.01D0 * STATE(S))
    CASE DEFAULT
       REPRESSILATOR_VARIANT_PROP = PARAMETERS(1) / (1D0 + PARAMETERS(2) * STATE(R)**6.0D0)
       IF (STATE(3) == 30 .AND. S == 1) REPRESSILATOR_VARIANT_PROP = REPRESSILATOR_VARIANT_PROP + 1.0D-3
    END SELECT
  END FUNCTION REPRESSILATOR_VARIANT_PROP

  DOUBLE PRECISION FUNCTION GOUTSIAS_EXAMPLE_PROP(STATE, REACTION, PARAMETERS)
    ! the propensities of examples/transcr6d.f90:63-90, restated; species M, D, RNA, DNA, DNA.D, DNA.2D = 1..6
    INTEGER, INTENT(IN) :: STATE(:), REACTION
    DOUBLE PRECISION, INTENT(IN), OPTIONAL :: PARAMETERS(:)
    INTEGER, PARAMETER :: FIRST(10) = [3, 1, 5, 3, 4, 5, 5, 6, 1, 2]
    SELECT CASE (REACTION)
    CASE (5, 7)
       GOUTSIAS_EXAMPLE_PROP = PARAMETERS(REACTION) * STATE(FIRST(REACTION)) * STATE(2)
    CASE (9)
       GOUTSIAS_EXAMPLE_PROP = PARAMETERS(9) * (STATE(1) * (STATE(1) - 1) / 2)
    CASE DEFAULT
       GOUTSIAS_EXAMPLE_PROP = PARAMETERS(REACTION) * STATE(FIRST(REACTION))
    END SELECT
  END FUNCTION GOUTSIAS_EXAMPLE_PROP

  SUBROUTINE LOAD_INPUT_MODEL(NAME, MODEL, X0)
    ! the three complete models/*.input files (keywords upper-cased by the
    ! Makefile: ModelModule.f90:95-140 matches upper case only) with the
    ! parameter values the reference's own drivers use.
    CHARACTER(LEN=*), INTENT(IN) :: NAME
    TYPE(CME_MODEL), INTENT(INOUT) :: MODEL
    INTEGER, ALLOCATABLE, INTENT(OUT) :: X0(:)
    SELECT CASE (NAME)
    CASE ('toggle')
       CALL MODEL%LOAD('toggle_model.input')
       ! test/TestSolverFromFile.f90:31
       CALL MODEL%RESET_PARAMETERS([1.0D0, 100.0D0, 1.0D0, 1.0D0, 100.0D0, 1.0D0])
       X0 = [0, 0]
    CASE ('repressilator')
       CALL MODEL%LOAD('repressilator_model.input')
       CALL MODEL%RESET_PARAMETERS([100.0D0, 100.0D0, 100.0D0, 1.0D0, 1.0D0, 1.0D0])
       X0 = [22, 0, 0]          ! examples/repressilator.f90:36
    CASE ('goutsias')
       CALL MODEL%LOAD('goutsias_model.input')
       ! examples/transcr6d.f90:23-32
       CALL MODEL%RESET_PARAMETERS([0.043D0, 0.0007D0, 0.0715D0, 0.0039D0, &
            0.0199264663575241D0, 0.4791D0, 0.000199264663575241D0, &
            0.8765D0 * 1.0D-11, 0.0830269431563506104D0, 0.5D0])
       X0 = [2, 6, 0, 2, 0, 0]  ! examples/transcr6d.f90:50
    CASE DEFAULT
       STOP 'ref_cases: unknown model'
    END SELECT
  END SUBROUTINE LOAD_INPUT_MODEL

  ! capacity of the FSPs the solve cases create: TABLEN, or the (prime) number in the environment
  ! variable KFSP_CASE_CAPACITY for runs that outgrow it (Goutsias at T = 300 reaches 1.03e6 states;
  ! 2097169 and 6291469 = NMAX, StateSpace.f90:10, are primes)
  INTEGER FUNCTION CASE_CAPACITY()
    CHARACTER(LEN=32) :: ENV
    INTEGER :: L, STAT
    CASE_CAPACITY = TABLEN
    CALL GET_ENVIRONMENT_VARIABLE('KFSP_CASE_CAPACITY', ENV, L, STAT)
    IF (STAT == 0 .AND. L > 0) READ(ENV(1:L), *, IOSTAT=STAT) CASE_CAPACITY
    IF (CASE_CAPACITY < 16) CASE_CAPACITY = TABLEN
  END FUNCTION CASE_CAPACITY

  SUBROUTINE SEED_POINT(MODEL, FSP_IN, FSP, X0)
    TYPE(CME_MODEL), INTENT(IN) :: MODEL
    TYPE(FINITE_STATE_PROJECTION), INTENT(INOUT) :: FSP_IN, FSP
    INTEGER, INTENT(IN) :: X0(:)
    CALL FSP_IN%CREATE(MODEL, CASE_CAPACITY())
    CALL FSP%CREATE(MODEL, CASE_CAPACITY())
    FSP_IN%SIZE = 1
    FSP_IN%STATE(:, 1) = X0
    FSP_IN%VECTOR = 0.0D0
    FSP_IN%VECTOR(1) = 1.0D0
    FSP = FSP_IN                  ! as the drivers do (toggle.f90:45)
  END SUBROUTINE SEED_POINT

  SUBROUTINE SEED_RING(MODEL, FSP_IN, FSP, NMOL)
    TYPE(CME_MODEL), INTENT(INOUT) :: MODEL
    TYPE(FINITE_STATE_PROJECTION), INTENT(INOUT) :: FSP_IN, FSP
    INTEGER, INTENT(IN) :: NMOL
    INTEGER :: NS, I, K, N, X(6), S
    DOUBLE PRECISION :: TOT
    NS = RING_NS
    CALL MODEL%CREATE(NS, 2 * NS, 2 * NS)
    MODEL%STOICHIOMETRY = 0
    DO I = 1, NS
       K = MOD(I, NS) + 1
       ! reaction 2i-1 : S_i -> S_{i+1} ; reaction 2i : S_{i+1} -> S_i
       MODEL%STOICHIOMETRY(I, 2 * I - 1) = -1
       MODEL%STOICHIOMETRY(K, 2 * I - 1) = 1
       MODEL%STOICHIOMETRY(K, 2 * I) = -1
       MODEL%STOICHIOMETRY(I, 2 * I) = 1
    ENDDO
    MODEL%CUSTOMPROP => RING_PROP
    MODEL%PARAMETER_VAL = 0.0D0
    MODEL%LOADED = .TRUE.
    CALL FSP_IN%CREATE(MODEL, TABLEN)
    CALL FSP%CREATE(MODEL, TABLEN)
    ! enumerate every composition of NMOL into NS parts (odometer on the
    ! first NS-1 coordinates)
    N = 0
    X = 0
    DO
       S = SUM(X(1:NS - 1))
       IF (S <= NMOL) THEN
          N = N + 1
          FSP_IN%STATE(1:NS - 1, N) = X(1:NS - 1)
          FSP_IN%STATE(NS, N) = NMOL - S
       ENDIF
       I = 1
       DO WHILE (I <= NS - 1)
          X(I) = X(I) + 1
          IF (X(I) <= NMOL) EXIT
          X(I) = 0
          I = I + 1
       ENDDO
       IF (I > NS - 1) EXIT
    ENDDO
    FSP_IN%SIZE = N
    FSP_IN%VECTOR = 0.0D0
    TOT = 0.0D0
    DO I = 1, N
       FSP_IN%VECTOR(I) = 1.0D0 + 0.5D0 * SIN(DBLE(I))
       TOT = TOT + FSP_IN%VECTOR(I)
    ENDDO
    FSP_IN%VECTOR(1:N) = FSP_IN%VECTOR(1:N) / TOT
    FSP = FSP_IN
  END SUBROUTINE SEED_RING

  ! the named CME_SOLVE workloads: model, seed FSP, horizon and tolerances
  SUBROUTINE SETUP_SOLVE_CASE(CASENAME, MODEL, FSP_IN, FSP, T, FSPTOL, KRYTOL)
    CHARACTER(LEN=*), INTENT(IN) :: CASENAME
    TYPE(CME_MODEL), INTENT(INOUT) :: MODEL
    TYPE(FINITE_STATE_PROJECTION), INTENT(INOUT) :: FSP_IN, FSP
    DOUBLE PRECISION, INTENT(OUT) :: T, FSPTOL, KRYTOL
    INTEGER, ALLOCATABLE :: X0(:)
    INTEGER :: NMOL
    SELECT CASE (CASENAME)
    CASE ('toggle_input')
       ! test/TestSolverFromFile.f90:35
       CALL LOAD_INPUT_MODEL('toggle', MODEL, X0)
       T = 1000.0D0; FSPTOL = 1.0D-4; KRYTOL = 1.0D-10
       CALL SEED_POINT(MODEL, FSP_IN, FSP, X0)
    CASE ('toggle_example')
       ! examples/toggle.f90:14-48 (compiled-in propensities)
       CALL MODEL%CREATE(2, 4, 6)
       MODEL%STOICHIOMETRY = RESHAPE((/1, 0, -1, 0, 0, 1, 0, -1/), (/2, 4/))
       MODEL%CUSTOMPROP => TOGGLE_EXAMPLE_PROP
       CALL MODEL%RESET_PARAMETERS([1.0D0, 100.0D0, 1.0D0, 1.0D0, 100.0D0, 1.0D0])
       MODEL%LOADED = .TRUE.
       T = 100.0D0; FSPTOL = 1.0D-4; KRYTOL = 1.0D-8
       X0 = [0, 0]
       CALL SEED_POINT(MODEL, FSP_IN, FSP, X0)
    CASE ('repressilator_input')
       ! models/repressilator_model.input with the tolerances of toggle_input
       CALL LOAD_INPUT_MODEL('repressilator', MODEL, X0)
       T = 10.0D0; FSPTOL = 1.0D-4; KRYTOL = 1.0D-10
       CALL SEED_POINT(MODEL, FSP_IN, FSP, X0)
    CASE ('goutsias_input')
       ! models/goutsias_model.input with the tolerances of examples/transcr6d.f90:16
       CALL LOAD_INPUT_MODEL('goutsias', MODEL, X0)
       T = 300.0D0; FSPTOL = 1.0D-6; KRYTOL = 1.0D-8
       CALL SEED_POINT(MODEL, FSP_IN, FSP, X0)
    CASE ('repressilator_example')
       ! examples/repressilator.f90:14-42 (compiled-in propensities; stoichiometry :24, parameters :26, seed :36)
       CALL MODEL%CREATE(3, 6, 3)
       MODEL%STOICHIOMETRY = RESHAPE((/1, 0, 0, -1, 0, 0, 0, 1, 0, 0, -1, 0, 0, 0, 1, 0, 0, -1/), (/3, 6/))
       MODEL%CUSTOMPROP => REPRESSILATOR_EXAMPLE_PROP
       CALL MODEL%RESET_PARAMETERS([100.0D0, 25.0D0, 1.0D0])
       MODEL%LOADED = .TRUE.
       T = 10.0D0; FSPTOL = 1.0D-4; KRYTOL = 1.0D-14
       X0 = [22, 0, 0]
       CALL SEED_POINT(MODEL, FSP_IN, FSP, X0)
    CASE ('repressilator_pair', 'repressilator_triple', 'repressilator_trap')
       CALL MODEL%CREATE(3, 6, 4)
       MODEL%STOICHIOMETRY = RESHAPE((/1, 0, 0, -1, 0, 0, 0, 1, 0, 0, -1, 0, 0, 0, 1, 0, 0, -1/), (/3, 6/))
       MODEL%CUSTOMPROP => REPRESSILATOR_VARIANT_PROP
       CALL MODEL%RESET_PARAMETERS([100.0D0, 25.0D0, 1.0D0, 1.0D0])
       IF (CASENAME == 'repressilator_triple') MODEL%PARAMETER_VAL(4) = 2.0D0
       IF (CASENAME == 'repressilator_trap') MODEL%PARAMETER_VAL(4) = 3.0D0
       MODEL%LOADED = .TRUE.
       T = 2.0D0; FSPTOL = 1.0D-4; KRYTOL = 1.0D-10
       X0 = [22, 0, 0]
       CALL SEED_POINT(MODEL, FSP_IN, FSP, X0)
    CASE ('goutsias_example')
       ! examples/transcr6d.f90:14-56 (compiled-in propensities :63-90, stoichiometry :92-131, parameters :23-32)
       CALL MODEL%CREATE(6, 10, 10)
       MODEL%STOICHIOMETRY = 0
       MODEL%STOICHIOMETRY(1, 1) = 1;   MODEL%STOICHIOMETRY(1, 2) = -1
       MODEL%STOICHIOMETRY(3, 3) = 1;   MODEL%STOICHIOMETRY(3, 4) = -1
       MODEL%STOICHIOMETRY([4, 2, 5], 5) = [-1, -1, 1];  MODEL%STOICHIOMETRY([4, 2, 5], 6) = [1, 1, -1]
       MODEL%STOICHIOMETRY([5, 2, 6], 7) = [-1, -1, 1];  MODEL%STOICHIOMETRY([5, 2, 6], 8) = [1, 1, -1]
       MODEL%STOICHIOMETRY([1, 2], 9) = [-2, 1];         MODEL%STOICHIOMETRY([1, 2], 10) = [2, -1]
       MODEL%CUSTOMPROP => GOUTSIAS_EXAMPLE_PROP
       CALL MODEL%RESET_PARAMETERS([0.043D0, 0.0007D0, 0.0715D0, 0.0039D0, &
            0.0199264663575241D0, 0.4791D0, 0.000199264663575241D0, &
            0.8765D0 * 1.0D-11, 0.0830269431563506104D0, 0.5D0])
       MODEL%LOADED = .TRUE.
       T = 300.0D0; FSPTOL = 1.0D-6; KRYTOL = 1.0D-8
       X0 = [2, 6, 0, 2, 0, 0]
       CALL SEED_POINT(MODEL, FSP_IN, FSP, X0)
    CASE ('ring6')
       ! closed 6-species ring, 8 molecules: every state is seeded, nothing
       ! leaks and nothing is droppable, so the FSP never changes and the run
       ! is a pure fixed-matrix adaptive expv (pins KrylovSolver.f90:206-550
       ! with :509-534 idle).
       RING_NS = 6; NMOL = 8
       RING_CF = [1.00D0, 1.10D0, 0.90D0, 1.05D0, 0.95D0, 1.00D0]
       RING_CB = [0.90D0, 1.00D0, 1.10D0, 0.95D0, 1.05D0, 1.00D0]
       T = 60.0D0; FSPTOL = 1.0D-4; KRYTOL = 1.0D-10
       CALL SEED_RING(MODEL, FSP_IN, FSP, NMOL)
    CASE ('ring4')
       RING_NS = 4; NMOL = 12
       RING_CF = [2.00D0, 1.50D0, 1.80D0, 2.20D0, 0.0D0, 0.0D0]
       RING_CB = [1.70D0, 2.10D0, 1.60D0, 1.90D0, 0.0D0, 0.0D0]
       T = 50.0D0; FSPTOL = 1.0D-6; KRYTOL = 1.0D-8
       CALL SEED_RING(MODEL, FSP_IN, FSP, NMOL)
    CASE DEFAULT
       STOP 'ref_cases: unknown solve case'
    END SELECT
    ! (scaling experiments: KFSP_CASE_FSPTOL overrides the FSP tolerance of the named case - a tighter one grows a larger FSP)
    BLOCK
      CHARACTER(LEN=32) :: ENV
      INTEGER :: L, STAT
      DOUBLE PRECISION :: V
      CALL GET_ENVIRONMENT_VARIABLE('KFSP_CASE_FSPTOL', ENV, L, STAT)
      IF (STAT == 0 .AND. L > 0) THEN
         READ(ENV(1:L), *, IOSTAT=STAT) V
         IF (STAT == 0 .AND. V > 0.0D0) FSPTOL = V
      ENDIF
    END BLOCK
  END SUBROUTINE SETUP_SOLVE_CASE

  ! fsp file : int32 ns, nr, n ; int32 STATE(ns,n) ; int32 ADJ(nr,n) ;
  !            f64 OFFDIAG(nr,n) ; f64 DIAG(n) ; f64 VECTOR(n)
  SUBROUTINE WRITE_FSP(FNAME, MODEL, FSP)
    CHARACTER(LEN=*), INTENT(IN) :: FNAME
    TYPE(CME_MODEL), INTENT(IN) :: MODEL
    TYPE(FINITE_STATE_PROJECTION), INTENT(IN) :: FSP
    INTEGER :: U, N
    N = FSP%SIZE
    OPEN(NEWUNIT=U, FILE=FNAME, ACCESS='STREAM', FORM='UNFORMATTED', STATUS='REPLACE')
    WRITE(U) MODEL%NSPECIES, MODEL%NREACTIONS, N
    WRITE(U) FSP%STATE(1:MODEL%NSPECIES, 1:N)
    WRITE(U) FSP%MATRIX%ADJ(1:MODEL%NREACTIONS, 1:N)
    WRITE(U) FSP%MATRIX%OFFDIAG(1:MODEL%NREACTIONS, 1:N)
    WRITE(U) FSP%MATRIX%DIAG(1:N)
    WRITE(U) FSP%VECTOR(1:N)
    CLOSE(U)
  END SUBROUTINE WRITE_FSP
END MODULE REF_CASES
