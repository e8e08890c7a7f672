! TEST INFRASTRUCTURE - NOT PRODUCT CODE.
!
! kfsp_replay: CME_SOLVE of OUR modules (krylovfspssa_amd/fortran) on one of the
! workloads of oracle/ref_cases.f90, in lock step with a run of the unmodified
! reference: the decision script made by oracle/make_golden.py from a ref_trace
! run steers kfsp_dgexpv_replay, everything else - the device arithmetic, our
! DROP_STATES / SSA_EXTENDER / ONESTEP_EXTENDER on OUR vector - runs as in a free
! solve.  The observer writes the state list and the solution vector after every
! step in the record layout of oracle/ref_trace.c ('B' / 'F'), so that the test
! reads both runs with one reader and compares them step by step.
!
! usage:  kfsp_replay <case> <script.bin> <steps.bin> <out.bin> [T] [safe|safestop|-] [digest]
!         kfsp_replay rkey <model> <out.txt>     COMPUTE_RKEY of our STATESPACE for one of the
!                                                .input models: one line "sign key" per reaction
!         kfsp_replay proptable <out.bin> [notab]   the grid of ref_dump exprtable (expr_test_model.input, 13 x 13 x 3
!                                                states, 16 propensities) evaluated ON THE DEVICE through the model's
!                                                propensity program (kfsp_propensities); notab: no host-made tables,
!                                                every expression through the device's interpreter
!   script.bin: int64 rows; f64 script(4, rows)
!   forks go to <steps.bin>.forks: int32 rc, nforks, rows_used_lo, rows_used_hi ... see below
MODULE REPLAY_OBSERVER
  USE, INTRINSIC :: ISO_C_BINDING
  USE STATESPACE
  IMPLICIT NONE
  INTEGER :: OUT_UNIT = -1, ONS = 0, ONR = 0
  LOGICAL :: DIGEST = .FALSE.
  INTEGER, ALLOCATABLE :: LAST_STATE(:, :)
CONTAINS
  SUBROUTINE OBSERVE(NSTEP, T_NOW, BETA, FSP)
    INTEGER, INTENT(IN) :: NSTEP
    DOUBLE PRECISION, INTENT(IN) :: T_NOW, BETA
    TYPE(FINITE_STATE_PROJECTION), INTENT(IN) :: FSP
    INTEGER :: N
    LOGICAL :: SAME
    N = FSP%SIZE
    IF (DIGEST) THEN
       CALL WRITE_DIGEST(N, BETA, FSP)
       RETURN
    ENDIF
    WRITE(OUT_UNIT) 'B', INT(N, C_INT32_T), BETA, FSP%VECTOR(1:N)
    SAME = .FALSE.
    IF (ALLOCATED(LAST_STATE)) THEN
       IF (SIZE(LAST_STATE, 2) == N) SAME = ALL(LAST_STATE == FSP%STATE(1:ONS, 1:N))
    ENDIF
    IF (.NOT. SAME) THEN
       LAST_STATE = FSP%STATE(1:ONS, 1:N)
       WRITE(OUT_UNIT) 'F', INT(ONS, C_INT32_T), INT(ONR, C_INT32_T), INT(N, C_INT32_T)
       WRITE(OUT_UNIT) FSP%STATE(1:ONS, 1:N)
       WRITE(OUT_UNIT) FSP%MATRIX%ADJ(1:ONR, 1:N)
       WRITE(OUT_UNIT) FSP%MATRIX%OFFDIAG(1:ONR, 1:N)
       WRITE(OUT_UNIT) FSP%MATRIX%DIAG(1:N)
    ENDIF
  END SUBROUTINE OBSERVE

  ! digest mode (argument 7 = 'digest'; runs whose lists are too big to write out): record 'D' = size,
  ! beta, the checksum pair of the state list and eight weighted sums of the vector, formed exactly
  ! as oracle/lockstep.py list_hash / state_weights form them for the reference's run
  SUBROUTINE WRITE_DIGEST(N, BETA, FSP)
    INTEGER, INTENT(IN) :: N
    DOUBLE PRECISION, INTENT(IN) :: BETA
    TYPE(FINITE_STATE_PROJECTION), INTENT(IN) :: FSP
    INTEGER(8), PARAMETER :: MULT(8) = [1000003_8, 998244353_8, 19260817_8, 1000000007_8, 74207281_8, &
         433494437_8, 2971215073_8, 32452843_8]
    INTEGER(8), PARAMETER :: M = 2147483647_8
    INTEGER(8) :: H, A, B, I8
    DOUBLE PRECISION :: PROJ(8), C(8), X
    INTEGER :: I, S, J
    DO J = 1, 8
       C(J) = 0.0001D0 * DBLE(J) + 1.0D-9 * DBLE(J)**2
    ENDDO
    A = 0_8
    B = 0_8
    PROJ = 0.0D0
    !$OMP PARALLEL DO SCHEDULE(STATIC) PRIVATE(H, S, I8, X, J) REDUCTION(+:A, B, PROJ) IF(N > 20000)
    DO I = 1, N
       H = 0_8
       DO S = 1, ONS
          H = H + INT(FSP%STATE(S, I), 8) * MULT(S)
       ENDDO
       H = MOD(H, M)
       I8 = INT(I, 8)
       A = A + MOD((MOD(I8, 1000003_8) + 1_8) * H, M)
       B = B + MOD((MOD(I8, 999983_8) + 7_8) * MOD(H * 48271_8 + 11_8, M), M)
       X = DBLE(H)
       DO J = 1, 8
          PROJ(J) = PROJ(J) + FSP%VECTOR(I) * COS(X * C(J))
       ENDDO
    ENDDO
    !$OMP END PARALLEL DO
    WRITE(OUT_UNIT) 'D', INT(N, C_INT32_T), BETA, MOD(A, M), MOD(B, M), PROJ
  END SUBROUTINE WRITE_DIGEST
END MODULE REPLAY_OBSERVER

PROGRAM KFSP_REPLAY_MAIN
  USE, INTRINSIC :: ISO_C_BINDING
  USE REF_CASES
  USE REPLAY_OBSERVER
  USE STATESPACE
  USE KRYLOVSOLVER
  IMPLICIT NONE
  CHARACTER(LEN=256) :: CASENAME, SCRIPTFILE, STEPFILE, OUTFILE, TARG, SARG, DARG
  TYPE(CME_MODEL) :: MODEL
  TYPE(FINITE_STATE_PROJECTION) :: FSP_IN, FSP
  DOUBLE PRECISION :: T, FSPTOL, KRYTOL
  REAL(C_DOUBLE), ALLOCATABLE, TARGET :: SCRIPT(:, :)
  INTEGER(C_INT64_T) :: ROWS
  INTEGER :: U, I, NF

  CALL GET_COMMAND_ARGUMENT(1, CASENAME)
  CALL GET_COMMAND_ARGUMENT(2, SCRIPTFILE)
  CALL GET_COMMAND_ARGUMENT(3, STEPFILE)
  CALL GET_COMMAND_ARGUMENT(4, OUTFILE)
  CALL GET_COMMAND_ARGUMENT(5, TARG)
  CALL GET_COMMAND_ARGUMENT(6, SARG)
  CALL GET_COMMAND_ARGUMENT(7, DARG)
  IF (TRIM(CASENAME) == 'rkey') THEN
     CALL DO_RKEY(TRIM(SCRIPTFILE), TRIM(STEPFILE))
     STOP
  ENDIF
  IF (TRIM(CASENAME) == 'proptable') THEN
     CALL DO_DEVICE_PROPTABLE(TRIM(SCRIPTFILE), TRIM(STEPFILE))
     STOP
  ENDIF
  CALL RANDOM_SEED()
  CALL SETUP_SOLVE_CASE(TRIM(CASENAME), MODEL, FSP_IN, FSP, T, FSPTOL, KRYTOL)
  IF (LEN_TRIM(TARG) > 0 .AND. TRIM(TARG) /= '-') READ(TARG, *) T
  IF (TRIM(SARG) == 'safe') KFSP_REPLAY_SAFE = 1
  IF (TRIM(SARG) == 'safestop') KFSP_REPLAY_SAFE = 2     ! ... and stop where the state lists part
  DIGEST = TRIM(DARG) == 'digest'

  IF (TRIM(SCRIPTFILE) /= '-') THEN
     OPEN(NEWUNIT=U, FILE=TRIM(SCRIPTFILE), ACCESS='STREAM', FORM='UNFORMATTED', STATUS='OLD')
     READ(U) ROWS
     ALLOCATE(SCRIPT(4, ROWS))
     READ(U) SCRIPT
     CLOSE(U)
     KFSP_REPLAY_SCRIPT => SCRIPT
  ENDIF
  ONS = MODEL%NSPECIES
  ONR = MODEL%NREACTIONS
  OPEN(NEWUNIT=OUT_UNIT, FILE=TRIM(STEPFILE), ACCESS='STREAM', FORM='UNFORMATTED', STATUS='REPLACE')
  KFSP_STEP_OBSERVER => OBSERVE
  CALL CME_SOLVE(MODEL, T, FSP_IN, FSP, FSPTOL, KRYTOL, VERBOSITY = 1)
  CLOSE(OUT_UNIT)
  CALL WRITE_FSP(TRIM(OUTFILE), MODEL, FSP)

  ! what the lock step found: rc, number of differences, max |WSUM - recorded WSUM|, then the
  ! entries (step, kind, own(2), forced(2), lhs, rhs)
  NF = MIN(KFSP_REPLAY_NFORKS, 4096)
  OPEN(NEWUNIT=U, FILE=TRIM(STEPFILE) // '.forks', ACCESS='STREAM', FORM='UNFORMATTED', STATUS='REPLACE')
  WRITE(U) INT(KFSP_REPLAY_RC, C_INT32_T), INT(KFSP_REPLAY_NFORKS, C_INT32_T), KFSP_REPLAY_WSUM_DIFF, &
       INT(KFSP_REPLAY_EXTENSIONS, C_INT32_T), 0_C_INT32_T
  DO I = 1, NF
     WRITE(U) KFSP_REPLAY_FORKS(I)%STEP, KFSP_REPLAY_FORKS(I)%KIND, KFSP_REPLAY_FORKS(I)%OWN, &
          KFSP_REPLAY_FORKS(I)%FORCED, KFSP_REPLAY_FORKS(I)%LHS, KFSP_REPLAY_FORKS(I)%RHS
  ENDDO
  CLOSE(U)
  PRINT *, 'REPLAY RC =', KFSP_REPLAY_RC, ' DIFFERENCES =', KFSP_REPLAY_NFORKS, ' MAX WSUM DIFF =', KFSP_REPLAY_WSUM_DIFF, &
       ' BASIS EXTENSIONS =', KFSP_REPLAY_EXTENSIONS
  DO I = 1, MIN(NF, 40)
     PRINT '(A,I5,A,I2,A,2ES24.16,A,2ES24.16,A,2ES24.16)', ' FORK STEP', KFSP_REPLAY_FORKS(I)%STEP, ' KIND', &
          KFSP_REPLAY_FORKS(I)%KIND, ' OWN', KFSP_REPLAY_FORKS(I)%OWN, ' FORCED', KFSP_REPLAY_FORKS(I)%FORCED, &
          ' LHS/RHS', KFSP_REPLAY_FORKS(I)%LHS, KFSP_REPLAY_FORKS(I)%RHS
  ENDDO
  PRINT *, 'FINAL SIZE', FSP%SIZE
  PRINT *, 'FINAL SUM ', SUM(FSP%VECTOR(1:FSP%SIZE))
CONTAINS
  SUBROUTINE DO_DEVICE_PROPTABLE(FNAME, MODE)
    CHARACTER(LEN=*), INTENT(IN) :: FNAME, MODE
    TYPE(CME_MODEL) :: M
    INTEGER :: ST(3, 3 * 13 * 13), I, J, K, N, UU
    DOUBLE PRECISION :: OFF(16, 3 * 13 * 13), DG(3 * 13 * 13)
    CALL M%LOAD('expr_test_model.input')
    CALL M%RESET_PARAMETERS((/7.5D0, 2.0D0, 0.75D0, 0.3D0, 4.0D0/))
    N = 0
    DO I = 0, 12                                   ! the order of ref_dump's P(16, 3, 13, 13): DNA.2D fastest, then Y, X
       DO J = 0, 12
          DO K = 0, 2
             N = N + 1
             ST(:, N) = (/I, J, K/)
          ENDDO
       ENDDO
    ENDDO
    CALL KFSP_UPLOAD_PROGRAM(M, NO_TABLES=(MODE == 'notab'))
    IF (.NOT. KFSP_DEVICE_PROPENSITIES(N, ST, OFF, DG)) STOP 'no propensity program on the device'
    OPEN(NEWUNIT=UU, FILE=FNAME, ACCESS='STREAM', FORM='UNFORMATTED', STATUS='REPLACE')
    WRITE(UU) OFF
    WRITE(UU) DG
    CLOSE(UU)
  END SUBROUTINE DO_DEVICE_PROPTABLE

  SUBROUTINE DO_RKEY(NAME, FNAME)
    CHARACTER(LEN=*), INTENT(IN) :: NAME, FNAME
    TYPE(CME_MODEL) :: M
    INTEGER, ALLOCATABLE :: X0(:), SG(:)
    INTEGER(16), ALLOCATABLE :: RK(:)
    INTEGER :: J, UU
    CALL LOAD_INPUT_MODEL(NAME, M, X0)
    ALLOCATE(RK(M%NREACTIONS), SG(M%NREACTIONS))
    CALL COMPUTE_RKEY(RK, SG, M%NSPECIES, M%NREACTIONS, M)
    OPEN(NEWUNIT=UU, FILE=FNAME, STATUS='REPLACE')
    DO J = 1, M%NREACTIONS
       WRITE(UU, '(I3,1X,I40)') SG(J), RK(J)
    ENDDO
    CLOSE(UU)
  END SUBROUTINE DO_RKEY
END PROGRAM KFSP_REPLAY_MAIN
