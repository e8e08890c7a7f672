! MODULE KRYLOVSOLVER - the solver entry points of the reference
! (src/fsp/KrylovSolver.f90:7-36 CME_SOLVE, :40-653 DGEXPV_FSP) on top of the
! MI355X hot path.
!
! The Fortran side owns what the reference's host code owns - the model, the
! state list, the generator columns and the decisions about growing and
! shrinking the FSP - and hands the time loop to libkfsp_hip through the C ABI
! (module KFSP_C, include/kfsp.h):
!     generator columns  -> kfsp_set_matrix_ell   (FSP%MATRIX verbatim)
!     probability vector -> kfsp_set_vector / kfsp_get_vector
!     time loop          -> kfsp_dgexpv, which calls back here when the FSP has
!                           to change: CB_DROP (DROP_STATES, :509-512) and
!                           CB_EXPAND (SSA_EXTENDER + ONESTEP_EXTENDER, :518-534)
!     log lines          -> CB_LOG prints what the reference prints
! There is no CPU fallback: without the library / a GPU the solve stops.
MODULE KRYLOVSOLVER
  USE, INTRINSIC :: ISO_C_BINDING
  USE STATESPACE
  USE KFSP_CUSTOMPROP
  USE KFSP_C
  IMPLICIT NONE

  ! statistics of the last solve (the reference computes them into local
  ! arrays and loses them, :554-573)
  TYPE(KFSP_STATS), SAVE :: LAST_SOLVE_STATS

  ! Observer (optional): called whenever w and the FSP are final for the step that
  ! begins next - once before the first step (KrylovSolver.f90:177) and after every
  ! step's DROP_STATES / SSA expansion (:540).  FSP%VECTOR(1:FSP%SIZE) holds w.
  ABSTRACT INTERFACE
     SUBROUTINE KFSP_OBSERVER(NSTEP, T_NOW, BETA, FSP)
       IMPORT :: FINITE_STATE_PROJECTION
       INTEGER, INTENT(IN) :: NSTEP
       DOUBLE PRECISION, INTENT(IN) :: T_NOW, BETA
       TYPE(FINITE_STATE_PROJECTION), INTENT(IN) :: FSP
     END SUBROUTINE KFSP_OBSERVER
  END INTERFACE
  PROCEDURE(KFSP_OBSERVER), POINTER, SAVE :: KFSP_STEP_OBSERVER => NULL()

  ! Lock-step diagnostics (kfsp_dgexpv_replay, include/kfsp.h): when a script is
  ! associated the next solve follows it; the differences found are left in
  ! KFSP_REPLAY_FORKS(1:MIN(KFSP_REPLAY_NFORKS, SIZE)) and KFSP_REPLAY_RC (0, or 20
  ! when the script and the run went out of step).
  REAL(C_DOUBLE), POINTER, SAVE :: KFSP_REPLAY_SCRIPT(:, :) => NULL()     ! (4, rows)
  TYPE(KFSP_FORK), ALLOCATABLE, TARGET, SAVE :: KFSP_REPLAY_FORKS(:)
  INTEGER, SAVE :: KFSP_REPLAY_NFORKS = 0, KFSP_REPLAY_RC = 0
  INTEGER, SAVE :: KFSP_REPLAY_SAFE = 0          ! kfsp_replay.safe for the next solve
  INTEGER, SAVE :: KFSP_REPLAY_EXTENSIONS = 0    ! kfsp_replay.n_safe_extensions of the last one
  DOUBLE PRECISION, SAVE :: KFSP_REPLAY_WSUM_DIFF = 0.0D0

  TYPE(C_PTR), SAVE, PRIVATE :: CTX = C_NULL_PTR
  TYPE(FINITE_STATE_PROJECTION), POINTER, SAVE, PRIVATE :: CUR_FSP => NULL()
  TYPE(CME_MODEL), POINTER, SAVE, PRIVATE :: CUR_MODEL => NULL()
  INTEGER, SAVE, PRIVATE :: CUR_TRACE = 0
  LOGICAL, SAVE, PRIVATE :: HOST_DROP = .FALSE.      ! KFSP_HOST_DROP=1: DROP_STATES decided on the host
  ! the model's propensity program is resident on the device (kfsp_set_propensity_program): one-step sweeps on the
  ! device return the complete columns of the states they append.  KFSP_DEVICE_PROPENSITY=0 keeps them on the host.
  LOGICAL, SAVE, PRIVATE :: PROGRAM_READY = .FALSE., PROGRAM_WANTED = .TRUE.
  ! Resident mode (KFSP_RESIDENT=0 turns it off): with independent-stream SSA paths (KFSP_SSA_STREAMS=1) and the model's
  ! program on one device, DROP_STATES and the expansion step run on the device's own copy of the FSP
  ! (kfsp_drop_rebuild / kfsp_expand_resident); this side keeps the size only and fetches the lists when the solve is over.
  LOGICAL, SAVE, PRIVATE :: RESIDENT_WANTED = .TRUE., RESIDENT = .FALSE., SSA_ON_DEVICE = .TRUE.
  LOGICAL, SAVE, PRIVATE :: DEVICE_REBUILD = .TRUE.   ! KFSP_DEVICE_REBUILD=0: the compacted generator is uploaded after every drop
  ! A compiled-in CUSTOMPROP on the device (module KFSP_CUSTOMPROP): the plan the probe made, whether the program on the
  ! device IS that plan (then the solve is speculative: every propensity of the final lists is checked against the function,
  ! and the solve is repeated with host propensities if one differs), whether the tables could not follow the FSP
  ! (CUSTOM_FAILED: same repeat), and how often they grew.  KFSP_DEVICE_CUSTOMPROP=0: never probe.
  TYPE(CUSTOM_PLAN), SAVE, PRIVATE :: CPLAN
  LOGICAL, SAVE, PRIVATE :: CUSTOM_WANTED = .TRUE., CUSTOM_ACTIVE = .FALSE., CUSTOM_FAILED = .FALSE.
  INTEGER, SAVE, PRIVATE :: CUSTOM_GROWTHS = 0
  LOGICAL, SAVE, PRIVATE :: DEBUG_LINKS = .FALSE.    ! KFSP_DEBUG_LINKS=1: size and largest link after every host-side expansion sweep
  ! wall seconds spent in the host state-space code of the current solve:
  ! (1) DROP_STATES decision + compaction, (2) SSA_EXTENDER, (3) ONESTEP_EXTENDER,
  ! (4) uploads of the changed FSP
  DOUBLE PRECISION, SAVE, PRIVATE :: HOST_SEC(4) = 0.0D0

  PRIVATE :: ENSURE_CONTEXT, UPLOAD_FSP, CHECK, CB_DROP, CB_EXPAND, CB_LOG, WALL, DEVICE_ONESTEP, DEVICE_SSA, SEND_CUSTOM, GROW_CUSTOM
  PUBLIC :: KFSP_UPLOAD_PROGRAM, KFSP_DEVICE_PROPENSITIES

CONTAINS

  SUBROUTINE CME_SOLVE(MODEL, T, FSP_IN, FSP_OUT, FSPTOL, EXP_TOL, VERBOSITY)
    TYPE(CME_MODEL), INTENT(IN) :: MODEL
    DOUBLE PRECISION, INTENT(IN) :: T
    TYPE(FINITE_STATE_PROJECTION) :: FSP_IN
    DOUBLE PRECISION, INTENT(IN) :: FSPTOL
    DOUBLE PRECISION, INTENT(IN) :: EXP_TOL
    TYPE(FINITE_STATE_PROJECTION) :: FSP_OUT
    INTEGER, INTENT(IN), OPTIONAL :: VERBOSITY
    INTEGER :: ITRACE, IFLAG
    DOUBLE PRECISION :: KRYTOL
    ITRACE = 0
    IF (PRESENT(VERBOSITY)) ITRACE = VERBOSITY
    KRYTOL = EXP_TOL
    PRINT *, 'CALLING DGEXPV_FSP'
    CALL DGEXPV_FSP(MODEL, T, FSP_IN%VECTOR, FSP_OUT, FSP_OUT%VECTOR, FSPTOL, KRYTOL, ITRACE, IFLAG)
  END SUBROUTINE CME_SOLVE

  ! p(T) = exp(T A) p(0) on an adaptively grown / pruned FSP.
  ! V: start vector on the seed states FSP%STATE(:,1:FSP%SIZE); on return
  ! FSP holds the final state list and FSP%VECTOR(1:FSP%SIZE) (= W) the result.
  SUBROUTINE DGEXPV_FSP(MODEL, T, V, FSP, W, FSPTOL, KRYTOL, ITRACE, IFLAG)
    TYPE(CME_MODEL), TARGET :: MODEL
    DOUBLE PRECISION, INTENT(IN) :: T
    DOUBLE PRECISION, INTENT(IN) :: V(:)
    DOUBLE PRECISION, TARGET :: W(:)
    DOUBLE PRECISION, INTENT(IN) :: FSPTOL
    DOUBLE PRECISION :: KRYTOL
    TYPE(FINITE_STATE_PROJECTION), TARGET :: FSP
    INTEGER :: ITRACE
    INTEGER :: IFLAG
    TYPE(KFSP_FSP_OPS) :: OPS
    TYPE(KFSP_REPLAY) :: RP
    DOUBLE PRECISION, ALLOCATABLE :: P0(:)
    REAL(C_DOUBLE) :: TMS(7)
    INTEGER, ALLOCATABLE :: SEED(:, :)
    INTEGER :: I, N0, RC, NFIN, NBAD
    INTEGER(8) :: C0, C1, CRATE
    LOGICAL :: WAS_RESIDENT
    DOUBLE PRECISION :: T0V
    REAL(C_DOUBLE) :: PADE_SEC(4)
    INTEGER(C_INT64_T) :: BUILD_INFO(6)

    IFLAG = 0
    ! the device context comes first: the threads its runtime starts must not
    ! inherit the one-core affinity the host sweeps give the calling thread
    CALL ENSURE_CONTEXT()
    N0 = FSP%SIZE
    ALLOCATE(P0(N0), SEED(SIZE(FSP%STATE, 1), N0))
    P0 = V(1:N0)                       ! DCOPY(FSP%SIZE, V, 1, W, 1)  :176
    SEED = FSP%STATE(:, 1:N0)
    CALL KFSP_UPLOAD_PROGRAM(MODEL, SEEDS=SEED)
    CALL SYSTEM_CLOCK(C0, CRATE)
    RC = KFSP_GET_TIMERS(CTX, TMS, 1_C_INT)
    CALL KFSP_PADM_PROFILE(PADE_SEC, 1_C_INT)
    HOST_SEC = 0.0D0

    ATTEMPT: DO
    ! the FSP the first step runs on (:130-134)
    CALL MATRIX_STARTER(FSP, MODEL)
    DO I = 1, 5
       CALL ONESTEP_EXTENDER(FSP, MODEL)
    ENDDO
    FSP%VECTOR(1:N0) = P0
    IF (FSP%SIZE > N0) FSP%VECTOR(N0 + 1:FSP%SIZE) = 0.0D0

    CUR_FSP => FSP
    CUR_MODEL => MODEL
    CUR_TRACE = ITRACE
    CUSTOM_FAILED = .FALSE.
    RESIDENT = RESIDENT_WANTED .AND. PROGRAM_READY .AND. SSA_ON_DEVICE .AND. SSA_STREAMS_REQUESTED() &
         .AND. .NOT. HOST_DROP .AND. DEVICE_REBUILD .AND. .NOT. ASSOCIATED(KFSP_STEP_OBSERVER) &
         .AND. .NOT. ASSOCIATED(KFSP_REPLAY_SCRIPT)
    RC = KFSP_SET_OPTION(CTX, 'keep_coords' // C_NULL_CHAR, INT(MERGE(1, 0, RESIDENT), C_INT64_T))
    CALL UPLOAD_FSP(FSP, MODEL)

    OPS%USER = C_NULL_PTR
    OPS%DROP = C_FUNLOC(CB_DROP)
    OPS%EXPAND = C_FUNLOC(CB_EXPAND)
    OPS%LOG = C_FUNLOC(CB_LOG)
    IF (ASSOCIATED(KFSP_REPLAY_SCRIPT)) THEN
       IF (.NOT. ALLOCATED(KFSP_REPLAY_FORKS)) ALLOCATE(KFSP_REPLAY_FORKS(4096))
       RP%SCRIPT = C_LOC(KFSP_REPLAY_SCRIPT(1, 1))
       RP%N_ROWS = SIZE(KFSP_REPLAY_SCRIPT, 2)
       RP%FORKS = C_LOC(KFSP_REPLAY_FORKS(1))
       RP%MAX_FORKS = SIZE(KFSP_REPLAY_FORKS)
       RP%SAFE = KFSP_REPLAY_SAFE
       RC = KFSP_DGEXPV_REPLAY(CTX, T, FSPTOL, KRYTOL, INT(MODEL%NREACTIONS, C_INT), OPS, LAST_SOLVE_STATS, RP)
       KFSP_REPLAY_NFORKS = RP%N_FORKS
       KFSP_REPLAY_WSUM_DIFF = RP%MAX_WSUM_DIFF
       KFSP_REPLAY_EXTENSIONS = RP%N_SAFE_EXTENSIONS
       KFSP_REPLAY_RC = RC
       IF (RC == 20 .OR. RC == 21) RC = 0
       IF (.NOT. (CUSTOM_ACTIVE .AND. CUSTOM_FAILED)) CALL CHECK(RC, 'kfsp_dgexpv_replay')
    ELSE
       RC = KFSP_DGEXPV(CTX, T, FSPTOL, KRYTOL, INT(MODEL%NREACTIONS, C_INT), OPS, LAST_SOLVE_STATS)
       IF (.NOT. (CUSTOM_ACTIVE .AND. CUSTOM_FAILED)) CALL CHECK(RC, 'kfsp_dgexpv')
    ENDIF

    WAS_RESIDENT = RESIDENT
    IF (RESIDENT .AND. .NOT. CUSTOM_FAILED) THEN
       ! the lists as the device left them; keys and look-up table of this side follow
       NFIN = FSP%SIZE
       RC = KFSP_DOWNLOAD_FSP(CTX, INT(FSP%SIZE, C_INT32_T), FSP%STATE, INT(SIZE(FSP%STATE, 1), C_INT32_T), FSP%MATRIX%ADJ, &
            FSP%MATRIX%OFFDIAG, INT(SIZE(FSP%MATRIX%ADJ, 1), C_INT32_T), FSP%MATRIX%DIAG)
       CALL CHECK(RC, 'kfsp_download_fsp')
       CALL ADOPT_LISTS(FSP, MODEL, NFIN)
    ENDIF
    RESIDENT = .FALSE.
    IF (.NOT. CUSTOM_ACTIVE) EXIT ATTEMPT
    ! The program on the device was made by PROBING a compiled-in function: the result stands only if every propensity
    ! of the final lists is the function's own value, bit for bit.  Otherwise (or when the tables could not follow the
    ! FSP) the solve is repeated from the seed with the propensities - and with them the lists - on the host.
    NBAD = 0
    IF (.NOT. CUSTOM_FAILED) THEN
       T0V = WALL()
       NBAD = CUSTOM_VERIFY(MODEL, FSP)
       IF (ITRACE /= 0) PRINT '(A,I10,A,I4,A,F9.1,A,I8)', ' KFSP CUSTOMPROP ON THE DEVICE: ', FSP%SIZE * (MODEL%NREACTIONS + 1), &
            ' PROPENSITIES VERIFIED AGAINST THE FUNCTION, TABLE GROWTHS =', CUSTOM_GROWTHS, ', MS =', 1.0D3 * (WALL() - T0V), &
            ', MISMATCHES =', NBAD
    ENDIF
    IF (NBAD == 0 .AND. .NOT. CUSTOM_FAILED) EXIT ATTEMPT
    PRINT *, 'KFSP: THE DEVICE TABLES OF CUSTOMPROP DID NOT HOLD (MISMATCHES =', NBAD, ', TABLE LIMIT =', CUSTOM_FAILED, &
         '); REPEATING THE SOLVE WITH HOST PROPENSITIES'
    PROGRAM_READY = .FALSE.
    CUSTOM_ACTIVE = .FALSE.
    NULLIFY(SSA_DEVICE)
    FSP%SIZE = N0
    FSP%MATRIX%SIZE = N0
    FSP%STATE(:, 1:N0) = SEED
    ENDDO ATTEMPT

    RC = KFSP_GET_VECTOR(CTX, INT(FSP%SIZE, C_INT64_T), FSP%VECTOR)
    CALL CHECK(RC, 'kfsp_get_vector')
    ! W is FSP%VECTOR itself when called through CME_SOLVE; a distinct W gets a copy
    IF (.NOT. C_ASSOCIATED(C_LOC(W(1)), C_LOC(FSP%VECTOR(1)))) THEN
       IF (SIZE(W) >= FSP%SIZE) W(1:FSP%SIZE) = FSP%VECTOR(1:FSP%SIZE)
    ENDIF
    NULLIFY(CUR_FSP, CUR_MODEL)
    IF (ITRACE /= 0) THEN
       ! where the wall time of the solve went (ms); FSP_CALLBACKS = host state-space
       ! code (DROP_STATES / SSA_EXTENDER / ONESTEP_EXTENDER) incl. its uploads
       CALL SYSTEM_CLOCK(C1)
       RC = KFSP_GET_TIMERS(CTX, TMS, 0_C_INT)
       PRINT '(A,F12.1,A,7(1X,A,F11.1))', ' KFSP WALL MS =', 1.0D3 * DBLE(C1 - C0) / DBLE(CRATE), ' :', &
            'ARNOLDI', TMS(1), 'COMBINE', TMS(2), 'BEGIN_STEP', TMS(3), 'FSP_CALLBACKS', TMS(4), &
            'HOST_PADE', TMS(5), 'UPLOAD', TMS(6), 'DEVICE_ONESTEP', TMS(7)
       ! which side kept the lists (RESIDENT: the device, DESIGN.md 10.6), whether the model's propensities were there, and
       ! how the SSA paths were sampled (REFERENCE: one stream, the reference's order; STREAMS: one stream per path)
       PRINT '(A,A,A,A,A,A,A,A)', ' KFSP MODE: LISTS = ', TRIM(MERGE('RESIDENT', 'HOST    ', WAS_RESIDENT)), ' PROPENSITIES = ', &
            TRIM(MERGE('DEVICE', 'HOST  ', PROGRAM_READY)), ' SSA = ', TRIM(MERGE('STREAMS  ', 'REFERENCE', SSA_STREAMS_REQUESTED())), &
            ' MODEL = ', TRIM(MERGE('CUSTOMPROP ', 'EXPRESSIONS', ASSOCIATED(MODEL%CUSTOMPROP)))
       ! (rebuilds of the resident FSP that did not stop for their sizes, and those repeated because a check failed: kfsp.h)
       IF (WAS_RESIDENT .AND. KFSP_BUILD_INFO(CTX, BUILD_INFO) == 0) &
            PRINT '(A,I7,A,I5,A,I7)', ' KFSP RESIDENT REBUILDS: SPECULATIVE =', BUILD_INFO(1), ' REPEATED =', BUILD_INFO(2), &
            ' ORDERS CARRIED OVER =', BUILD_INFO(5)
       CALL KFSP_PADM_PROFILE(PADE_SEC, 1_C_INT)
       PRINT '(A,4(1X,A,F9.1))', ' KFSP HOST PADE PARTS MS:', 'DENSE_PRODUCTS', 1.0D3 * PADE_SEC(1), 'BANDED_PRODUCTS', 1.0D3 * PADE_SEC(2), &
            'SOLVE', 1.0D3 * PADE_SEC(3), 'WHOLE_CALLS', 1.0D3 * PADE_SEC(4)
       PRINT '(A,I8,A,I8,A,I8,A,I6,A,I6)', ' KFSP STATS: NMULT =', LAST_SOLVE_STATS%NMULT, ' NEXPH =', &
            LAST_SOLVE_STATS%NEXPH, ' WSUM_EVALS =', LAST_SOLVE_STATS%N_WSUM, ' EXPANSIONS =', &
            LAST_SOLVE_STATS%N_EXPAND, ' DROP_CALLS =', LAST_SOLVE_STATS%N_DROP_CALLS
       PRINT '(A,4(1X,A,F10.1))', ' KFSP HOST STATE-SPACE MS:', 'DROP_STATES', 1.0D3 * HOST_SEC(1), &
            'SSA_EXTENDER', 1.0D3 * HOST_SEC(2), 'ONESTEP_EXTENDER', 1.0D3 * HOST_SEC(3), 'UPLOADS', 1.0D3 * HOST_SEC(4)
       ! the passes inside them (process totals): ONESTEP scan/append/link,
       ! SSA walk/link, DROP flags/compact/renumber/table
       PRINT '(A,9F9.1)', ' KFSP HOST PASSES MS:', 1.0D3 * STATESPACE_SEC
    ENDIF
  END SUBROUTINE DGEXPV_FSP

  ! ------------------------------------------------------------- plumbing

  DOUBLE PRECISION FUNCTION WALL()
    INTEGER(8) :: C, R
    CALL SYSTEM_CLOCK(C, R)
    WALL = DBLE(C) / DBLE(R)
  END FUNCTION WALL

  SUBROUTINE ENSURE_CONTEXT()
    CHARACTER(LEN=16) :: ENV
    CHARACTER(LEN=512) :: OPTS
    INTEGER :: P0, P1, PE
    INTEGER :: DEV, L, STAT, RC, NRANKS
    INTEGER(C_INT) :: DEVS(64)
    INTEGER(C_INT64_T) :: V8
    IF (C_ASSOCIATED(CTX)) RETURN
    DEV = 0
    CALL GET_ENVIRONMENT_VARIABLE('KFSP_DEVICE', ENV, L, STAT)
    IF (STAT == 0 .AND. L > 0) READ(ENV(1:L), *, IOSTAT=STAT) DEV
    ! KFSP_NRANKS = P > 1: the FSP is row-partitioned over P contexts behind one head handle (a group context,
    ! include/kfsp.h); this program stays what it is - one thread, one copy of the state space.  KFSP_DEVICES
    ! = "d1,d2,..." names their devices (distinct devices: RCCL over xGMI); without it all P sit on KFSP_DEVICE
    ! and exchange through the loop-back transport (one-GPU rehearsal of the partition).
    NRANKS = 1
    CALL GET_ENVIRONMENT_VARIABLE('KFSP_NRANKS', ENV, L, STAT)
    IF (STAT == 0 .AND. L > 0) READ(ENV(1:L), *, IOSTAT=STAT) NRANKS
    IF (NRANKS > 1) THEN
       NRANKS = MIN(NRANKS, SIZE(DEVS))
       DEVS = DEV
       CALL GET_ENVIRONMENT_VARIABLE('KFSP_DEVICES', OPTS, L, STAT)
       IF (STAT == 0 .AND. L > 0) READ(OPTS(1:L), *, IOSTAT=STAT) DEVS(1:NRANKS)
       RC = KFSP_CREATE_GROUP(INT(NRANKS, C_INT), DEVS, CTX)
    ELSE
       RC = KFSP_CREATE(INT(DEV, C_INT), CTX)
    ENDIF
    IF (RC /= 0) THEN
       PRINT *, 'KFSP: NO USABLE HIP DEVICE (kfsp_create returned', RC, '); THE SOLVER HAS NO CPU PATH.'
       STOP 2
    ENDIF
    ! from now on ONESTEP_EXTENDER has a device for the integer work of its sweeps
    ONESTEP_DEVICE => DEVICE_ONESTEP
    ! The device keeps large, long-lived FSPs in its own (lexicographic) state order - ON by
    ! default.  Every row is still summed in FMATVEC's order (KrylovSolver.f90:598-604), so products
    ! are bit-identical to the plain path; the reductions over states (DNRM2 / DDOT / DASUM and the
    ! FIND_DROPTOL sums on the device) add their terms in the internal order.  KFSP_STATE_ORDER=0
    ! restores the caller's order everywhere; KFSP_STATE_ORDER_MIN: smallest FSP that is reordered
    ! (library default 32768), KFSP_STATE_ORDER_PRODUCTS: products the previous generator must
    ! have seen (default 48)
    CALL GET_ENVIRONMENT_VARIABLE('KFSP_STATE_ORDER_MIN', ENV, L, STAT)
    IF (STAT == 0 .AND. L > 0) THEN
       READ(ENV(1:L), *, IOSTAT=STAT) V8
       IF (STAT == 0) RC = KFSP_SET_OPTION(CTX, 'state_order_min' // C_NULL_CHAR, V8)
    ENDIF
    CALL GET_ENVIRONMENT_VARIABLE('KFSP_STATE_ORDER_PRODUCTS', ENV, L, STAT)
    IF (STAT == 0 .AND. L > 0) THEN
       READ(ENV(1:L), *, IOSTAT=STAT) V8
       IF (STAT == 0) RC = KFSP_SET_OPTION(CTX, 'state_order_products' // C_NULL_CHAR, V8)
    ENDIF
    CALL GET_ENVIRONMENT_VARIABLE('KFSP_STATE_ORDER', ENV, L, STAT)
    IF (STAT == 0 .AND. L > 0) THEN
       READ(ENV(1:L), *, IOSTAT=STAT) V8
       IF (STAT == 0) RC = KFSP_SET_OPTION(CTX, 'state_order' // C_NULL_CHAR, V8)
    ENDIF
    CALL GET_ENVIRONMENT_VARIABLE('KFSP_HOST_DROP', ENV, L, STAT)
    HOST_DROP = (STAT == 0 .AND. L > 0 .AND. ENV(1:1) /= '0')
    CALL GET_ENVIRONMENT_VARIABLE('KFSP_RESIDENT', ENV, L, STAT)
    IF (STAT == 0 .AND. L > 0) RESIDENT_WANTED = ENV(1:1) /= '0'
    CALL GET_ENVIRONMENT_VARIABLE('KFSP_DEVICE_SSA', ENV, L, STAT)
    IF (STAT == 0 .AND. L > 0) SSA_ON_DEVICE = ENV(1:1) /= '0'
    CALL GET_ENVIRONMENT_VARIABLE('KFSP_DEVICE_REBUILD', ENV, L, STAT)
    IF (STAT == 0 .AND. L > 0) DEVICE_REBUILD = ENV(1:1) /= '0'
    CALL GET_ENVIRONMENT_VARIABLE('KFSP_DEVICE_PROPENSITY', ENV, L, STAT)
    IF (STAT == 0 .AND. L > 0) PROGRAM_WANTED = ENV(1:1) /= '0'
    CALL GET_ENVIRONMENT_VARIABLE('KFSP_DEBUG_LINKS', ENV, L, STAT)
    DEBUG_LINKS = (STAT == 0 .AND. L > 0 .AND. ENV(1:1) /= '0')
    CALL GET_ENVIRONMENT_VARIABLE('KFSP_DEVICE_CUSTOMPROP', ENV, L, STAT)
    IF (STAT == 0 .AND. L > 0) CUSTOM_WANTED = ENV(1:1) /= '0' 
    ! any other library option (kfsp_set_option, include/kfsp.h): KFSP_OPTIONS="name=value,name=value"
    CALL GET_ENVIRONMENT_VARIABLE('KFSP_OPTIONS', OPTS, L, STAT)
    IF (STAT == 0 .AND. L > 0) THEN
       P0 = 1
       DO WHILE (P0 <= L)
          P1 = INDEX(OPTS(P0:L), ',')
          IF (P1 == 0) THEN
             P1 = L + 1
          ELSE
             P1 = P0 + P1 - 1
          ENDIF
          PE = INDEX(OPTS(P0:P1 - 1), '=')
          IF (PE > 1) THEN
             READ(OPTS(P0 + PE:P1 - 1), *, IOSTAT=STAT) V8
             IF (STAT == 0) THEN
                RC = KFSP_SET_OPTION(CTX, TRIM(ADJUSTL(OPTS(P0:P0 + PE - 2))) // C_NULL_CHAR, V8)
                IF (RC /= 0) PRINT *, 'KFSP: UNKNOWN OPTION IN KFSP_OPTIONS: ', OPTS(P0:P1 - 1)
             ENDIF
          ENDIF
          P0 = P1 + 1
       ENDDO
    ENDIF
  END SUBROUTINE ENSURE_CONTEXT

  SUBROUTINE CHECK(RC, WHAT)
    INTEGER(C_INT), INTENT(IN) :: RC
    CHARACTER(LEN=*), INTENT(IN) :: WHAT
    IF (RC == 0) RETURN
    PRINT *, 'KFSP: ', WHAT, ' FAILED WITH CODE ', RC, ': ', KFSP_ERROR_TEXT(CTX)
    STOP 3
  END SUBROUTINE CHECK

  ! STATESPACE's hook: kfsp_onestep on this module's context; the appended states and the
  ! completed links are written behind / into the caller's own arrays
  INTEGER FUNCTION DEVICE_ONESTEP(NS, NR, STOICH, N, STATE, ADJ, MAXCOUNT, CAP, NNEW, OFFDIAG, DIAG, COLUMNS)
    INTEGER, INTENT(IN) :: NS, NR, N, MAXCOUNT, CAP
    INTEGER, INTENT(IN) :: STOICH(NS, NR)
    INTEGER, INTENT(INOUT) :: STATE(NS, *), ADJ(NR, *)
    INTEGER, INTENT(OUT) :: NNEW
    DOUBLE PRECISION, INTENT(INOUT) :: OFFDIAG(NR, *), DIAG(*)
    LOGICAL, INTENT(OUT) :: COLUMNS
    INTEGER(C_INT32_T) :: NN
    NN = N
    COLUMNS = PROGRAM_READY
    IF (COLUMNS) THEN
       ! the propensity program of the model is on the device: complete columns come back
       DO
          DEVICE_ONESTEP = KFSP_ONESTEP_COLUMNS(CTX, INT(NS, C_INT32_T), INT(NR, C_INT32_T), STOICH, INT(N, C_INT32_T), STATE, &
               INT(NS, C_INT32_T), ADJ, INT(NR, C_INT32_T), INT(MAXCOUNT, C_INT32_T), INT(CAP, C_INT32_T), NN, &
               STATE(1:NS, N + 1:CAP), ADJ, OFFDIAG(1:NR, N + 1:CAP), INT(NR, C_INT32_T), DIAG(N + 1:CAP))
          ! (-16: a new state lies beyond a two-species table of a probed CUSTOMPROP - nothing was written; larger tables,
          ! again; when they cannot grow the host sweep takes over, here and from now on)
          IF (DEVICE_ONESTEP /= -16) EXIT
          IF (.NOT. GROW_CUSTOM()) EXIT
       ENDDO
    ELSE
       DEVICE_ONESTEP = KFSP_ONESTEP(CTX, INT(NS, C_INT32_T), INT(NR, C_INT32_T), STOICH, INT(N, C_INT32_T), STATE, &
            INT(NS, C_INT32_T), ADJ, INT(NR, C_INT32_T), INT(MAXCOUNT, C_INT32_T), INT(CAP, C_INT32_T), NN, &
            STATE(1:NS, N + 1:CAP), ADJ)
    ENDIF
    NNEW = NN
  END FUNCTION DEVICE_ONESTEP

  ! STATESPACE's hook for the independent-stream SSA walk: kfsp_ssa_streams on this module's context
  INTEGER FUNCTION DEVICE_SSA(TIMESTEP, SEEDMIX, NS, NR, STOICH, N, STATE, ADJ, OFFDIAG, DIAG, MAXCOUNT, CAPNEW, NFOUND)
    DOUBLE PRECISION, INTENT(IN) :: TIMESTEP
    INTEGER(8), INTENT(IN) :: SEEDMIX
    INTEGER, INTENT(IN) :: NS, NR, N, MAXCOUNT, CAPNEW
    INTEGER, INTENT(IN) :: STOICH(NS, NR)
    INTEGER, INTENT(INOUT) :: STATE(NS, *)
    INTEGER, INTENT(IN) :: ADJ(NR, *)
    DOUBLE PRECISION, INTENT(INOUT) :: OFFDIAG(NR, *), DIAG(*)
    INTEGER, INTENT(OUT) :: NFOUND
    INTEGER(C_INT32_T) :: NF
    NF = 0
    NFOUND = 0
    DEVICE_SSA = -1
    IF (.NOT. PROGRAM_READY .OR. CAPNEW < 1) RETURN
    DO
       DEVICE_SSA = KFSP_SSA_STREAMS(CTX, TIMESTEP, INT(SEEDMIX, C_INT64_T), INT(NS, C_INT32_T), INT(NR, C_INT32_T), STOICH, &
            INT(N, C_INT32_T), STATE, INT(NS, C_INT32_T), ADJ, OFFDIAG, INT(NR, C_INT32_T), DIAG, INT(MAXCOUNT, C_INT32_T), &
            INT(CAPNEW, C_INT32_T), NF, STATE(1:NS, N + 1:N + CAPNEW), OFFDIAG(1:NR, N + 1:N + CAPNEW), INT(NR, C_INT32_T), &
            DIAG(N + 1:N + CAPNEW))
       IF (DEVICE_SSA /= -16) EXIT               ! (-16: a path left a two-species table - larger tables, the same walk again)
       IF (.NOT. GROW_CUSTOM()) EXIT
    ENDDO
    NFOUND = NF
  END FUNCTION DEVICE_SSA

  ! The model's parsed propensities -> the device (kfsp_set_propensity_program).  Expressions of ONE species (every
  ! Hill function and x (x - 1) / 2 of the shipped models) travel as tables made HERE with MODEL%PROPENSITY at every
  ! population count 0..MAXNUMBERMOLECULES, so the device returns the host's own bits for them; the others as postfix
  ! code (exact for + - * /).  Nothing happens for a compiled-in CUSTOMPROP (no code to hand over).
  SUBROUTINE KFSP_UPLOAD_PROGRAM(MODEL, NO_TABLES, SEEDS)
    TYPE(CME_MODEL), INTENT(IN) :: MODEL
    LOGICAL, INTENT(IN), OPTIONAL :: NO_TABLES         ! (tests: everything through the device's interpreter)
    INTEGER, INTENT(IN), OPTIONAL :: SEEDS(:, :)       ! the seed states: where a compiled-in CUSTOMPROP is probed (KFSP_CUSTOMPROP)
    INTEGER, ALLOCATABLE :: CODE_OFF(:), CODE(:), IMM_OFF(:), DEP(:), X(:)
    DOUBLE PRECISION, ALLOCATABLE :: IMM(:), TAB(:, :)
    DOUBLE PRECISION :: PDUMMY(1)
    LOGICAL :: OK, TABLES
    INTEGER :: K, V, TL
    INTEGER(C_INT) :: RC
    PROGRAM_READY = .FALSE.
    CUSTOM_ACTIVE = .FALSE.
    CUSTOM_GROWTHS = 0
    NULLIFY(SSA_DEVICE)
    CALL ENSURE_CONTEXT()
    IF (.NOT. PROGRAM_WANTED) RETURN
    CALL MODEL%EXPORT_PROGRAM(OK, CODE_OFF, CODE, IMM_OFF, IMM, DEP)
    IF (.NOT. OK) THEN
       ! a compiled-in function: no code to hand over - probe it and tabulate it with the function itself (speculative:
       ! DGEXPV_FSP verifies the final lists and falls back to host propensities)
       IF (ASSOCIATED(MODEL%CUSTOMPROP) .AND. CUSTOM_WANTED .AND. PRESENT(SEEDS)) THEN
          IF (MODEL%NSPECIES > 16 .OR. MODEL%NREACTIONS > 64 .OR. SIZE(SEEDS, 2) < 1) RETURN
          CALL CUSTOM_PROBE(MODEL, SEEDS, SIZE(SEEDS, 2), CPLAN)
          IF (.NOT. CPLAN%OK) RETURN
          PROGRAM_READY = SEND_CUSTOM(MODEL) == 0
          CUSTOM_ACTIVE = PROGRAM_READY
          IF (PROGRAM_READY) SSA_DEVICE => DEVICE_SSA
       ENDIF
       RETURN
    ENDIF
    IF (MODEL%NSPECIES > 16 .OR. MODEL%NREACTIONS > 64) RETURN
    TABLES = .TRUE.
    IF (PRESENT(NO_TABLES)) TABLES = .NOT. NO_TABLES
    TL = 0
    IF (TABLES .AND. ANY(DEP >= 0)) TL = MAXNUMBERMOLECULES + 1
    ALLOCATE(TAB(MAX(TL, 1), MODEL%NREACTIONS), X(MODEL%NSPECIES))
    TAB = 0.0D0
    IF (TL > 0) THEN
       DO K = 1, MODEL%NREACTIONS
          IF (DEP(K) < 0) CYCLE
          X = 0
          DO V = 0, TL - 1
             X(DEP(K) + 1) = V
             TAB(V + 1, K) = MODEL%PROPENSITY(X, K)
          ENDDO
       ENDDO
    ELSE
       DEP = -1
    ENDIF
    PDUMMY = 0.0D0
    IF (MODEL%NPARAMETERS > 0) THEN
       RC = KFSP_SET_PROPENSITY_PROGRAM(CTX, INT(MODEL%NSPECIES, C_INT32_T), INT(MODEL%NREACTIONS, C_INT32_T), &
            INT(MODEL%NPARAMETERS, C_INT32_T), MODEL%PARAMETER_VAL, CODE_OFF, CODE, IMM_OFF, IMM, DEP, INT(TL, C_INT32_T), TAB)
    ELSE
       RC = KFSP_SET_PROPENSITY_PROGRAM(CTX, INT(MODEL%NSPECIES, C_INT32_T), INT(MODEL%NREACTIONS, C_INT32_T), &
            0_C_INT32_T, PDUMMY, CODE_OFF, CODE, IMM_OFF, IMM, DEP, INT(TL, C_INT32_T), TAB)
    ENDIF
    PROGRAM_READY = RC == 0               ! (a program the device cannot take - stack too deep - stays on the host)
    ! with the program on the device the independent-stream SSA paths (KFSP_SSA_STREAMS=1) can be walked there
    IF (PROGRAM_READY) THEN
       SSA_DEVICE => DEVICE_SSA
    ELSE
       NULLIFY(SSA_DEVICE)
    ENDIF
  END SUBROUTINE KFSP_UPLOAD_PROGRAM

  ! the plan of a probed CUSTOMPROP -> kfsp_set_propensity_program + kfsp_set_propensity_tables2
  INTEGER FUNCTION SEND_CUSTOM(MODEL) RESULT(RC)
    TYPE(CME_MODEL), INTENT(IN) :: MODEL
    INTEGER, ALLOCATABLE :: CODE_OFF(:), CODE(:), IMM_OFF(:), TS(:), T2S1(:), T2S2(:), T2N1(:), T2N2(:)
    INTEGER(8), ALLOCATABLE :: T2OFF(:)
    INTEGER(8) :: T2LEN
    INTEGER :: TL
    DOUBLE PRECISION, ALLOCATABLE :: IMM(:), TAB(:, :), TAB2(:)
    DOUBLE PRECISION :: PDUMMY(1)
    CALL CUSTOM_ARRAYS(MODEL, CPLAN, CODE_OFF, CODE, IMM_OFF, IMM, TS, TL, TAB, T2S1, T2S2, T2N1, T2N2, T2OFF, T2LEN, TAB2)
    PDUMMY = 0.0D0
    ! (the parameter values are inside the tables and the chains' constants: the program itself has no parameters)
    RC = KFSP_SET_PROPENSITY_PROGRAM(CTX, INT(MODEL%NSPECIES, C_INT32_T), INT(MODEL%NREACTIONS, C_INT32_T), 0_C_INT32_T, PDUMMY, &
         CODE_OFF, CODE, IMM_OFF, IMM, TS, INT(TL, C_INT32_T), TAB)
    IF (RC /= 0 .OR. T2LEN == 0) RETURN
    RC = KFSP_SET_PROPENSITY_TABLES2(CTX, INT(MODEL%NREACTIONS, C_INT32_T), T2S1, T2S2, T2N1, T2N2, T2OFF, INT(T2LEN, C_INT64_T), TAB2)
  END FUNCTION SEND_CUSTOM

  ! after a -16: the two-species tables grow past the population that missed them and go to the device again.
  ! .FALSE.: they cannot (size limit) - the device no longer serves this model's propensities
  LOGICAL FUNCTION GROW_CUSTOM() RESULT(GREW)
    INTEGER(C_INT32_T) :: MISSED(16)
    INTEGER(C_INT) :: RC
    GREW = .FALSE.
    IF (.NOT. CUSTOM_ACTIVE .OR. .NOT. ASSOCIATED(CUR_MODEL)) RETURN
    MISSED = 0
    RC = KFSP_PROPENSITY_OVERFLOW(CTX, INT(CUR_MODEL%NSPECIES, C_INT32_T), MISSED)
    IF (RC == 0) THEN
       IF (CUSTOM_GROW(CPLAN, INT(MISSED(1:CUR_MODEL%NSPECIES)))) GREW = SEND_CUSTOM(CUR_MODEL) == 0
    ENDIF
    IF (CUR_TRACE /= 0) PRINT '(A,16I6)', ' KFSP CUSTOMPROP: A POPULATION LEFT A TWO-SPECIES TABLE; LARGEST MISSES PER SPECIES =', &
         MISSED(1:CUR_MODEL%NSPECIES)
    IF (GREW) THEN
       CUSTOM_GROWTHS = CUSTOM_GROWTHS + 1
    ELSE
       CUSTOM_FAILED = .TRUE.
       PROGRAM_READY = .FALSE.
       NULLIFY(SSA_DEVICE)
    ENDIF
  END FUNCTION GROW_CUSTOM

  ! OFFDIAG / DIAG columns of N states through the resident program (kfsp_propensities); .FALSE. without one
  LOGICAL FUNCTION KFSP_DEVICE_PROPENSITIES(N, STATE, OFFDIAG, DIAG) RESULT(DONE)
    INTEGER, INTENT(IN) :: N
    INTEGER, INTENT(IN) :: STATE(:, :)
    DOUBLE PRECISION, INTENT(OUT) :: OFFDIAG(:, :), DIAG(:)
    INTEGER(C_INT) :: RC
    DONE = .FALSE.
    IF (.NOT. PROGRAM_READY) RETURN
    DO
       RC = KFSP_PROPENSITIES(CTX, INT(N, C_INT32_T), STATE, INT(SIZE(STATE, 1), C_INT32_T), OFFDIAG, &
            INT(SIZE(OFFDIAG, 1), C_INT32_T), DIAG)
       IF (RC /= -16) EXIT
       IF (.NOT. GROW_CUSTOM()) RETURN
    ENDDO
    CALL CHECK(RC, 'kfsp_propensities')
    DONE = .TRUE.
  END FUNCTION KFSP_DEVICE_PROPENSITIES

  ! generator columns + probability vector of the current FSP -> device
  SUBROUTINE UPLOAD_FSP(FSP, MODEL, WITH_VECTOR, N_UNCHANGED)
    TYPE(FINITE_STATE_PROJECTION), INTENT(IN) :: FSP
    TYPE(CME_MODEL), INTENT(IN) :: MODEL
    LOGICAL, INTENT(IN), OPTIONAL :: WITH_VECTOR      ! .FALSE.: the device already holds the vector
    INTEGER, INTENT(IN), OPTIONAL :: N_UNCHANGED      ! leading states whose propensity columns the device still holds
    INTEGER(C_INT) :: RC
    INTEGER :: KEEP
    KEEP = 0
    IF (PRESENT(N_UNCHANGED)) KEEP = N_UNCHANGED
    ! the species counts let the device keep its own, locality-preserving state
    ! order if that was asked for (nothing changes on this side of the boundary)
    ! (after a growth step the coordinates of the first KEEP states are on the device already, like their propensity columns)
    RC = KFSP_UPDATE_STATE_COORDS(CTX, INT(FSP%SIZE, C_INT32_T), INT(MODEL%NSPECIES, C_INT32_T), &
         INT(SIZE(FSP%STATE, 1), C_INT32_T), FSP%STATE, INT(KEEP, C_INT32_T))
    CALL CHECK(RC, 'kfsp_update_state_coords')
    RC = KFSP_UPDATE_MATRIX_ELL(CTX, INT(FSP%SIZE, C_INT32_T), INT(MODEL%NREACTIONS, C_INT32_T), &
         INT(SIZE(FSP%MATRIX%ADJ, 1), C_INT32_T), FSP%MATRIX%ADJ, FSP%MATRIX%OFFDIAG, FSP%MATRIX%DIAG, INT(KEEP, C_INT32_T))
    CALL CHECK(RC, 'kfsp_update_matrix_ell')
    IF (PRESENT(WITH_VECTOR)) THEN
       IF (.NOT. WITH_VECTOR) RETURN
    ENDIF
    RC = KFSP_SET_VECTOR(CTX, INT(FSP%SIZE, C_INT64_T), FSP%VECTOR)
    CALL CHECK(RC, 'kfsp_set_vector')
  END SUBROUTINE UPLOAD_FSP

  ! DROP_STATES seam (StateSpace.f90:431-548).  The decision - FIND_DROPTOL's threshold sums, the
  ! marks, the derivative guard on A*w, DROP_COUNT and the 10 % rule - is taken on the device on
  ! the resident vector (kfsp_drop_plan); nothing moves unless a compaction is due.  Then the flags
  ! come over (one byte per state), the host compacts ITS lists with them, the device compacts w
  ! where it lives, and only the generator of the compacted FSP is uploaded.
  ! KFSP_HOST_DROP=1 keeps the round-1 path (w and A*w to the host, decision there).
  FUNCTION CB_DROP(USER, DSUM, N_NEW) BIND(C) RESULT(RC)
    TYPE(C_PTR), VALUE :: USER
    REAL(C_DOUBLE), VALUE :: DSUM
    INTEGER(C_INT64_T) :: N_NEW
    INTEGER(C_INT) :: RC
    DOUBLE PRECISION, ALLOCATABLE :: WLOC(:), AW(:)
    INTEGER(C_INT8_T), ALLOCATABLE :: FLAGS(:)
    DOUBLE PRECISION :: D, T0
    REAL(C_DOUBLE) :: DROPTOL
    INTEGER(C_INT64_T) :: CNT, NFLAG, NKEEP
    LOGICAL :: CHANGED
    INTEGER :: N
    N = CUR_FSP%SIZE
    IF (.NOT. HOST_DROP) THEN
       T0 = WALL()
       RC = KFSP_DROP_PLAN(CTX, DSUM, DROPTOL, CNT, NFLAG)
       IF (RC /= 0) RETURN
       HOST_SEC(1) = HOST_SEC(1) + (WALL() - T0)
       N_NEW = N
       IF (DBLE(CNT) * 1.0D0 / (DBLE(N) * 1.0D0) <= 0.1D0) RETURN          ! :497
       T0 = WALL()
       IF (RESIDENT) THEN
          ! the vector and the device's own lists are compacted where they are; this side only learns the new size
          RC = KFSP_DROP_COMPACT(CTX, NKEEP)
          IF (RC /= 0) RETURN
          RC = KFSP_DROP_REBUILD(CTX)
          CALL CHECK(RC, 'kfsp_drop_rebuild')
          CUR_FSP%SIZE = INT(NKEEP)
          CUR_FSP%MATRIX%SIZE = INT(NKEEP)
          HOST_SEC(1) = HOST_SEC(1) + (WALL() - T0)
          N_NEW = NKEEP
          RC = 0
          RETURN
       ENDIF
       ALLOCATE(FLAGS(N))
       RC = KFSP_DROP_FLAGS(CTX, INT(N, C_INT64_T), FLAGS)
       IF (RC /= 0) RETURN
       RC = KFSP_DROP_COMPACT(CTX, NKEEP)
       IF (RC /= 0) RETURN
       CALL DROP_APPLY_FLAGS(CUR_FSP, CUR_MODEL, FLAGS)
       HOST_SEC(1) = HOST_SEC(1) + (WALL() - T0)
       IF (CUR_FSP%SIZE /= NKEEP) THEN
          RC = 4000
          RETURN
       ENDIF
       T0 = WALL()
       ! the device renumbers its own copy of the generator (kfsp_drop_rebuild): nothing travels; only when that
       ! copy is not there (-9) does the compacted FSP go up again
       RC = -9
       IF (DEVICE_REBUILD) RC = KFSP_DROP_REBUILD(CTX)
       IF (RC == -9) THEN
          CALL UPLOAD_FSP(CUR_FSP, CUR_MODEL, .FALSE.)
       ELSE
          CALL CHECK(RC, 'kfsp_drop_rebuild')
       ENDIF
       HOST_SEC(4) = HOST_SEC(4) + (WALL() - T0)
       N_NEW = CUR_FSP%SIZE
       RC = 0
       RETURN
    ENDIF
    ALLOCATE(WLOC(N), AW(N))
    RC = KFSP_GET_VECTOR(CTX, INT(N, C_INT64_T), WLOC)
    IF (RC /= 0) RETURN
    RC = KFSP_SPMV_W(CTX, AW)
    IF (RC /= 0) RETURN
    D = DSUM
    T0 = WALL()
    CALL DROP_STATES_CORE(WLOC, CUR_FSP, CUR_MODEL, D, AW, CHANGED)
    HOST_SEC(1) = HOST_SEC(1) + (WALL() - T0)
    IF (CHANGED) THEN
       CUR_FSP%VECTOR(1:N) = WLOC
       T0 = WALL()
       CALL UPLOAD_FSP(CUR_FSP, CUR_MODEL)
       HOST_SEC(4) = HOST_SEC(4) + (WALL() - T0)
    ENDIF
    N_NEW = CUR_FSP%SIZE
    RC = 0
  END FUNCTION CB_DROP

  ! expansion seam: SSA paths of length T_SSA from every state, then one sweep
  ! of one-step reachability; new states start with probability 0
  FUNCTION CB_EXPAND(USER, T_SSA, N_NEW) BIND(C) RESULT(RC)
    TYPE(C_PTR), VALUE :: USER
    REAL(C_DOUBLE), VALUE :: T_SSA
    INTEGER(C_INT64_T) :: N_NEW
    INTEGER(C_INT) :: RC
    DOUBLE PRECISION :: TS, T0, T1, T2, BASE
    INTEGER :: N_BEFORE
    INTEGER(C_INT64_T) :: NN, NSSA
    IF (RESIDENT) THEN
       ! SSA_EXTENDER_STREAMS + ONESTEP_EXTENDER on the device's own lists (kfsp_expand_resident); the number that seeds
       ! the paths' streams is drawn here, as SSA_EXTENDER_STREAMS draws it
       T0 = WALL()
       CALL RANDOM_NUMBER(BASE)
       DO
          RC = KFSP_EXPAND_RESIDENT(CTX, T_SSA, INT(BASE * 2147483647.0D0, C_INT64_T), INT(CUR_MODEL%NSPECIES, C_INT32_T), &
               INT(CUR_MODEL%NREACTIONS, C_INT32_T), CUR_MODEL%STOICHIOMETRY(1:CUR_MODEL%NSPECIES, 1:CUR_MODEL%NREACTIONS), &
               INT(MAXNUMBERMOLECULES, C_INT32_T), INT(CUR_FSP%MAX_SIZE - 1, C_INT32_T), NN, NSSA)
          ! (-16: a path or an appended state left a two-species table of a probed CUSTOMPROP; the lists are untouched -
          ! larger tables, the same step again)
          IF (RC /= -16) EXIT
          IF (.NOT. GROW_CUSTOM()) THEN
             RC = 4016                          ! (ends kfsp_dgexpv; DGEXPV_FSP repeats the solve with host propensities)
             RETURN
          ENDIF
       ENDDO
       IF (RC == -11) STOP 'OVERFLOW ERROR: FSP SIZE EXCEEDS MEMORY LIMIT.'
       CALL CHECK(RC, 'kfsp_expand_resident')
       CUR_FSP%SIZE = INT(NN)
       CUR_FSP%MATRIX%SIZE = INT(NN)
       HOST_SEC(2) = HOST_SEC(2) + (WALL() - T0)
       N_NEW = NN
       RC = 0
       RETURN
    ENDIF
    RC = KFSP_GET_VECTOR(CTX, INT(CUR_FSP%SIZE, C_INT64_T), CUR_FSP%VECTOR)
    IF (RC /= 0) RETURN
    N_BEFORE = CUR_FSP%SIZE
    TS = T_SSA
    T0 = WALL()
    ! (the FSP on the device IS the one the walk starts from: its arrays need not travel again)
    RC = KFSP_SET_OPTION(CTX, 'ssa_resident' // C_NULL_CHAR, 1_C_INT64_T)
    CALL SSA_EXTENDER(TS, CUR_FSP, CUR_MODEL)
    RC = KFSP_SET_OPTION(CTX, 'ssa_resident' // C_NULL_CHAR, 0_C_INT64_T)
    IF (DEBUG_LINKS) PRINT *, 'KFSP DEBUG: AFTER SSA_EXTENDER SIZE =', CUR_FSP%SIZE, ' MAX LINK =', &
         MAXVAL(CUR_FSP%MATRIX%ADJ(:, 1:CUR_FSP%SIZE))
    T1 = WALL()
    CALL ONESTEP_EXTENDER(CUR_FSP, CUR_MODEL)
    IF (DEBUG_LINKS) PRINT *, 'KFSP DEBUG: AFTER ONESTEP_EXTENDER SIZE =', CUR_FSP%SIZE, ' MAX LINK =', &
         MAXVAL(CUR_FSP%MATRIX%ADJ(:, 1:CUR_FSP%SIZE)), ' N_BEFORE =', N_BEFORE
    T2 = WALL()
    ! states only get appended here: the propensity columns of the first N_BEFORE are on the device already
    CALL UPLOAD_FSP(CUR_FSP, CUR_MODEL, N_UNCHANGED=N_BEFORE)
    HOST_SEC(2) = HOST_SEC(2) + (T1 - T0)
    HOST_SEC(3) = HOST_SEC(3) + (T2 - T1)
    HOST_SEC(4) = HOST_SEC(4) + (WALL() - T2)
    N_NEW = CUR_FSP%SIZE
    RC = 0
  END FUNCTION CB_EXPAND

  ! the reference's prints (KrylovSolver.f90:233-235, 384-389, 427-431, 452,
  ! 523-525, 641-651), same wording so that its logs and ours diff cleanly
  SUBROUTINE CB_LOG(USER, EVENT, VALS, NVALS) BIND(C)
    TYPE(C_PTR), VALUE :: USER
    INTEGER(C_INT), VALUE :: EVENT, NVALS
    REAL(C_DOUBLE), INTENT(IN) :: VALS(*)
    SELECT CASE (EVENT)
    CASE (KFSP_EV_WSUM)
       PRINT *, 'WSUM= ', VALS(1)
    CASE (KFSP_EV_BEGIN_IOP)
       IF (CUR_TRACE /= 0) PRINT *, 'BEGINNING IOP...'
    CASE (KFSP_EV_STEP)
       IF (CUR_TRACE /= 0) THEN
          PRINT *, 'TIMESTEP', INT(VALS(1)), '-------------------------------'
          PRINT *, 'FSP SIZE         =', INT(VALS(2))
          PRINT *, 'STEP_SIZE        =', VALS(3)
          PRINT *, 'NEXT_STEP        =', VALS(4)
          PRINT *, 'T_NOW            =', VALS(5)
          PRINT *, 'KRYLOV DIMENSION =', INT(VALS(6))
       ENDIF
    CASE (KFSP_EV_REJECT_STEP)
       IF (CUR_TRACE /= 0) THEN
          PRINT *, 'T_STEP =', VALS(1)
          PRINT *, 'ERR_LOC =', VALS(2)
          PRINT *, 'ERR_REQUIRED =', VALS(3)
          PRINT *, 'STEPSIZE REJECTED, DOWN TO:', VALS(4)
       ENDIF
    CASE (KFSP_EV_DIM_CHANGE)
       IF (CUR_TRACE /= 0) THEN
          PRINT *, 'ERR_LOC =', VALS(1)
          PRINT *, 'ERR_REQUIRED =', VALS(2)
          PRINT *, 'DIMENSION CHANGED INTO M =', INT(VALS(3))
       ENDIF
    CASE (KFSP_EV_CALL_SSA)
       IF (CUR_TRACE /= 0) PRINT *, 'CALLING SSA'
    CASE (KFSP_EV_READY)
       IF (ASSOCIATED(KFSP_STEP_OBSERVER)) THEN
          IF (KFSP_GET_VECTOR(CTX, INT(CUR_FSP%SIZE, C_INT64_T), CUR_FSP%VECTOR) == 0) &
               CALL KFSP_STEP_OBSERVER(INT(VALS(1)), VALS(2), VALS(3), CUR_FSP)
       ENDIF
    END SELECT
  END SUBROUTINE CB_LOG

END MODULE KRYLOVSOLVER
