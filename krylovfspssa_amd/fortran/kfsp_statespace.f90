! MODULE STATESPACE - the finite state projection: the ordered state list, the
! probability vector on it and the generator restricted to it in the column
! layout the solver uploads to the GPU (kfsp_set_matrix_ell takes it verbatim).
!
! Host-side drop-in surface with the reference's names and argument lists
! (src/state_space/StateSpace.f90): NMAX, MAXNUMBERMOLECULES, FSP_MATRIX,
! FINITE_STATE_PROJECTION (+ CREATE / CLEAR / PROBABILITY / ADD / INDEX),
! MATRIX_STARTER, ONESTEP_EXTENDER, SSA_EXTENDER, FIND_DROPTOL, DROP_STATES.
! The ORDER in which states enter and leave the list is kept exactly as in the
! reference (:136-246, :248-345, :347-396, :431-548, :550-630), because state
! indices are part of the results.  What is new: states are looked up through an
! open-addressing table keyed on the state vector itself (64-bit mixed hash +
! comparison of the stored state) instead of 140-byte big-integer keys with
! Brent's rehashing (HashTable.f90:61-236, big_integer_module.f90); only
! found / not-found and the stored index are observable, the table layout is not.
!
! Host parallelism.  A look-up is two or three dependent cache misses, and
! linking one new state needs 2*NREACTIONS of them, so the extenders are bound
! by memory latency.  They therefore split each sweep into (1) a sequential part
! that fixes the ORDER of the new states exactly as the reference's one-at-a-time
! ADD_STATE does (and calls the propensity functions in the same order), and
! (2) the linking of all new states afterwards, which only reads the table and
! runs under OpenMP.  Since every pair of listed states ends up linked either
! way, ADJ after the sweep is identical to the reference's.  KFSP_HOST_THREADS
! sets the thread count (default: all available, at most 16).
MODULE STATESPACE
  USE MODELMODULE
  USE, INTRINSIC :: ISO_C_BINDING, ONLY: C_INT, C_CHAR, C_NULL_CHAR
  !$ USE OMP_LIB
  IMPLICIT NONE

  ! capacity of a default-created FSP (prime in the reference; kept for the
  ! drivers' DIMENSION(NMAX) arrays)
  INTEGER, PARAMETER :: NMAX = 6291469
  INTEGER, PARAMETER :: MAXNUMBERMOLECULES = 10000

  TYPE FSP_MATRIX
     INTEGER :: SIZE = 0
     ! ADJ(k,i)  > 0 : index of state x_i + nu_k ; 0 : that state is outside the FSP ;
     !          -1 : it would have a negative population
     INTEGER, ALLOCATABLE :: ADJ(:, :)
     ! OFFDIAG(k,i) = a_k(x_i) ; DIAG(i) = sum_k a_k(x_i) (stored positive)
     DOUBLE PRECISION, ALLOCATABLE :: OFFDIAG(:, :), DIAG(:)
  END TYPE FSP_MATRIX

  TYPE :: FINITE_STATE_PROJECTION
     INTEGER :: MAX_SIZE = NMAX
     INTEGER :: SIZE = 0
     INTEGER, ALLOCATABLE :: STATE(:, :)
     ! 64-bit hash of every listed state (what the reference keeps as big-integer KEY)
     INTEGER(8), ALLOCATABLE :: KEY(:)
     TYPE(FSP_MATRIX) :: MATRIX
     DOUBLE PRECISION, ALLOCATABLE :: VECTOR(:)
     INTEGER, PRIVATE :: KTLEN = 0
     ! open-addressing table, one 8-byte word per slot: state index in the low
     ! half, 32 hash bits in the high half, 0 = free; it grows with the list
     ! (load factor <= 1/2).  KVTAB is kept as a name only (unused, length 1).
     INTEGER(8), ALLOCATABLE :: KEYTAB(:)
     INTEGER, ALLOCATABLE :: KVTAB(:)
   CONTAINS
     PROCEDURE :: CREATE => CREATE_FSP
     PROCEDURE :: CLEAR => CLEAR_FSP
     PROCEDURE :: PROBABILITY => POINTWISE_FSP
     PROCEDURE :: ADD => ADD_STATE
     PROCEDURE :: INDEX => INDEX_STATE
  END TYPE FINITE_STATE_PROJECTION

  PRIVATE :: ONESTEP_ON_DEVICE, SSA_STREAMS_ON_DEVICE, KFSP_PLOG, SSA_EXTENDER_STREAMS, STREAM_PATH, TICK, STATE_HASH, LOOKUP, PROBE, INSERT_RANGE, APPEND_CANDIDATES, TABLE_INSERT, REBUILD_TABLE, LEGAL, RESERVE_TABLE, APPEND_STATE, &
       LINK_ONE, LINK_NEW, HOST_THREADS

  INTEGER(8), PARAMETER, PRIVATE :: LOW32 = 4294967295_8

  INTERFACE
     ! setenv(3) of the C library
     INTEGER(C_INT) FUNCTION C_SETENV(NAME, VAL, OVERWRITE) BIND(C, NAME='setenv')
       IMPORT :: C_INT, C_CHAR
       CHARACTER(KIND=C_CHAR), INTENT(IN) :: NAME(*), VAL(*)
       INTEGER(C_INT), VALUE :: OVERWRITE
     END FUNCTION C_SETENV
  END INTERFACE
  PRIVATE :: C_SETENV
  INTEGER(8), PARAMETER, PRIVATE :: LCG_A = 48271_8, LCG_M = 2147483647_8, LCG_LOW = 1073741823_8
  DOUBLE PRECISION, PARAMETER, PRIVATE :: LCG_SCALE = 2.0D0**(-54)
  PRIVATE :: LCG_RECOGNISED

  ! wall seconds spent in the passes of the sweeps (profiles/statespace_bench.f90):
  ! 1-3 ONESTEP_EXTENDER scan / append / link, 4-5 SSA_EXTENDER walk / link,
  ! 6-9 DROP_STATES threshold+flags / compaction / renumbering / table
  DOUBLE PRECISION, SAVE :: STATESPACE_SEC(9) = 0.0D0

  ! ONESTEP_EXTENDER's integer work on the device (kfsp_onestep, include/kfsp.h): set by the solver
  ! module once it holds a device context; used for lists of at least ONESTEP_DEVICE_MIN states
  ! (environment KFSP_DEVICE_ONESTEP_MIN, default 20000; KFSP_DEVICE_ONESTEP=0 switches it off).
  ! Returns 0, -9 (more than 2^31 (state, reaction) pairs: the host sweep runs instead) or -11 (capacity).
  ABSTRACT INTERFACE
     INTEGER FUNCTION ONESTEP_DEVICE_FN(NS, NR, STOICH, N, STATE, ADJ, MAXCOUNT, CAP, NNEW, OFFDIAG, DIAG, COLUMNS)
       INTEGER, INTENT(IN) :: NS, NR, N, MAXCOUNT, CAP
       INTEGER, INTENT(IN) :: STOICH(NS, NR)
       INTEGER, INTENT(INOUT) :: STATE(NS, *), ADJ(NR, *)      ! new states / completed links are written in place
       INTEGER, INTENT(OUT) :: NNEW
       ! .TRUE. on return: OFFDIAG(:, N+1:NNEW) and DIAG(N+1:NNEW) were filled in as well (the model's propensity
       ! program runs on the device, kfsp_onestep_columns); .FALSE.: the caller evaluates the propensities
       DOUBLE PRECISION, INTENT(INOUT) :: OFFDIAG(NR, *), DIAG(*)
       LOGICAL, INTENT(OUT) :: COLUMNS
     END FUNCTION ONESTEP_DEVICE_FN
  END INTERFACE
  PROCEDURE(ONESTEP_DEVICE_FN), POINTER, SAVE :: ONESTEP_DEVICE => NULL()
  ! SSA_EXTENDER_STREAMS' walk on the device (kfsp_ssa_streams, include/kfsp.h): set by the solver module once the
  ! model's propensity program is on the device; used for lists of at least SSA_DEVICE_MIN states (environment
  ! KFSP_DEVICE_SSA_MIN, default 20000; KFSP_DEVICE_SSA=0 switches it off).  On return 0 the NFOUND distinct unlisted
  ! states the paths met stand behind the list - STATE(:, N+1:N+NFOUND) in (seed state, position) order of first
  ! occurrence - with their OFFDIAG / DIAG columns; any other value: the host walks.
  ABSTRACT INTERFACE
     INTEGER FUNCTION SSA_DEVICE_FN(TIMESTEP, SEEDMIX, NS, NR, STOICH, N, STATE, ADJ, OFFDIAG, DIAG, MAXCOUNT, CAPNEW, NFOUND)
       DOUBLE PRECISION, INTENT(IN) :: TIMESTEP
       INTEGER(8), INTENT(IN) :: SEEDMIX
       INTEGER, INTENT(IN) :: NS, NR, N, MAXCOUNT, CAPNEW
       INTEGER, INTENT(IN) :: STOICH(NS, NR)
       INTEGER, INTENT(INOUT) :: STATE(NS, *)
       INTEGER, INTENT(IN) :: ADJ(NR, *)
       DOUBLE PRECISION, INTENT(INOUT) :: OFFDIAG(NR, *), DIAG(*)
       INTEGER, INTENT(OUT) :: NFOUND
     END FUNCTION SSA_DEVICE_FN
  END INTERFACE
  PROCEDURE(SSA_DEVICE_FN), POINTER, SAVE :: SSA_DEVICE => NULL()
  INTEGER, SAVE, PRIVATE :: SSA_DEVICE_MIN = -1
  INTEGER, SAVE, PRIVATE :: ONESTEP_DEVICE_MIN = -1
  INTEGER, PRIVATE, SAVE :: NTHREADS_CACHED = 0, PARALLEL_MIN = -1, SSA_STREAMS_FLAG = -1
  INTEGER, PRIVATE, SAVE :: TOUCH_SINK = 0        ! keeps the early loads of SSA_EXTENDER alive

CONTAINS

  SUBROUTINE CREATE_FSP(FSP, MODEL, MAX_SIZE_CUSTOM)
    CLASS(FINITE_STATE_PROJECTION), INTENT(INOUT) :: FSP
    TYPE(CME_MODEL), INTENT(IN) :: MODEL
    INTEGER, OPTIONAL, INTENT(IN) :: MAX_SIZE_CUSTOM
    INTEGER :: N, M
    N = MODEL%NSPECIES
    M = MODEL%NREACTIONS
    IF (PRESENT(MAX_SIZE_CUSTOM)) FSP%MAX_SIZE = MAX_SIZE_CUSTOM
    IF (ALLOCATED(FSP%STATE)) CALL CLEAR_FSP(FSP)
    FSP%KTLEN = 1024
    ALLOCATE(FSP%STATE(N, FSP%MAX_SIZE), FSP%KEY(FSP%MAX_SIZE), &
         FSP%MATRIX%DIAG(FSP%MAX_SIZE), FSP%MATRIX%OFFDIAG(M, FSP%MAX_SIZE), FSP%MATRIX%ADJ(M, FSP%MAX_SIZE), &
         FSP%KEYTAB(FSP%KTLEN), FSP%KVTAB(1), FSP%VECTOR(FSP%MAX_SIZE))
    FSP%SIZE = 0
    FSP%MATRIX%SIZE = 0
    FSP%KEYTAB = 0_8
    FSP%KVTAB = 0
  END SUBROUTINE CREATE_FSP

  SUBROUTINE CLEAR_FSP(FSP)
    CLASS(FINITE_STATE_PROJECTION), INTENT(INOUT) :: FSP
    IF (ALLOCATED(FSP%STATE)) DEALLOCATE(FSP%STATE, FSP%KEY, FSP%MATRIX%DIAG, FSP%MATRIX%OFFDIAG, &
         FSP%MATRIX%ADJ, FSP%KEYTAB, FSP%KVTAB, FSP%VECTOR)
    FSP%SIZE = 0
    FSP%MATRIX%SIZE = 0
    FSP%KTLEN = 0
  END SUBROUTINE CLEAR_FSP

  ! adds the wall time since T0 to STATESPACE_SEC(SLOT) and restarts T0
  SUBROUTINE TICK(SLOT, T0)
    INTEGER, INTENT(IN) :: SLOT
    INTEGER(8), INTENT(INOUT) :: T0
    INTEGER(8) :: C, R
    CALL SYSTEM_CLOCK(C, R)
    IF (SLOT > 0) STATESPACE_SEC(SLOT) = STATESPACE_SEC(SLOT) + DBLE(C - T0) / DBLE(R)
    T0 = C
  END SUBROUTINE TICK

  ! threads for a sweep over WORK items that pays off from MINWORK items on
  ! (KFSP_HOST_PARALLEL_MIN replaces every MINWORK: the tests use it to run the
  ! threaded code on small cases)
  INTEGER FUNCTION HOST_THREADS(WORK, MINWORK)
    INTEGER, INTENT(IN) :: WORK, MINWORK
    CHARACTER(LEN=16) :: BUF
    INTEGER :: L, ST, V
    INTEGER(C_INT) :: RC
    IF (NTHREADS_CACHED == 0) THEN
       NTHREADS_CACHED = 1
       ! Keep the team on neighbouring cores of the caller's socket: what the
       ! sweeps write is read by the sequential parts right afterwards, and on a
       ! two-socket host an unbound team makes those reads remote.  The runtime
       ! reads its environment on first use, i.e. just below; values the user set
       ! are left alone.
       !$ RC = C_SETENV('OMP_PROC_BIND' // C_NULL_CHAR, 'close' // C_NULL_CHAR, 0_C_INT)
       !$ RC = C_SETENV('OMP_PLACES' // C_NULL_CHAR, 'cores' // C_NULL_CHAR, 0_C_INT)
       !$ NTHREADS_CACHED = MIN(OMP_GET_MAX_THREADS(), 16)
       ! the sweeps are milliseconds long and far apart: idle workers sleep
       ! instead of spinning (LLVM OpenMP runtime entry point)
       V = 0
       CALL GET_ENVIRONMENT_VARIABLE('KFSP_HOST_BLOCKTIME_MS', BUF, L, ST)
       IF (ST == 0 .AND. L > 0) READ(BUF(1:L), *, IOSTAT=ST) V
       !$ CALL KMP_SET_BLOCKTIME(MAX(V, 0))
       CALL GET_ENVIRONMENT_VARIABLE('KFSP_HOST_THREADS', BUF, L, ST)
       IF (ST == 0 .AND. L > 0) THEN
          READ(BUF(1:L), *, IOSTAT=ST) V
          IF (ST == 0 .AND. V >= 1) NTHREADS_CACHED = MIN(V, 256)
       ENDIF
       CALL GET_ENVIRONMENT_VARIABLE('KFSP_HOST_PARALLEL_MIN', BUF, L, ST)
       IF (ST == 0 .AND. L > 0) THEN
          READ(BUF(1:L), *, IOSTAT=ST) V
          IF (ST == 0 .AND. V >= 0) PARALLEL_MIN = V
       ENDIF
    ENDIF
    HOST_THREADS = 1
    IF (PARALLEL_MIN >= 0) THEN
       IF (WORK >= PARALLEL_MIN) HOST_THREADS = NTHREADS_CACHED
    ELSE
       IF (WORK >= MINWORK) HOST_THREADS = NTHREADS_CACHED
    ENDIF
  END FUNCTION HOST_THREADS

  ! ---------------------------------------------------------------- lookup

  PURE LOGICAL FUNCTION LEGAL(X)
    ! what STATE2KEY accepts (HashTable.f90:51-57)
    INTEGER, INTENT(IN) :: X(:)
    LEGAL = ALL(X >= 0) .AND. ALL(X <= MAXNUMBERMOLECULES)
  END FUNCTION LEGAL

  PURE INTEGER(8) FUNCTION STATE_HASH(X)
    ! multiplicative hash of the coordinates; H < 2**40 and the multiplier
    ! < 2**20, so the 64-bit product never overflows
    INTEGER, INTENT(IN) :: X(:)
    INTEGER(8), PARAMETER :: MASK40 = 1099511627775_8     ! 2**40 - 1
    INTEGER(8) :: H
    INTEGER :: K
    H = 1469598103_8
    DO K = 1, SIZE(X)
       H = IAND(H * 1000003_8 + INT(X(K), 8) + 1_8, MASK40)
       H = IEOR(H, ISHFT(H, -17))
    ENDDO
    STATE_HASH = H
  END FUNCTION STATE_HASH

  ! index of state X in the FSP (0 = absent) and its hash H.  The slot comes from
  ! the low bits of H, the tag kept beside the index from bits 8..39.
  SUBROUTINE LOOKUP(FSP, X, IDX, H)
    CLASS(FINITE_STATE_PROJECTION), INTENT(IN) :: FSP
    INTEGER, INTENT(IN) :: X(:)
    INTEGER, INTENT(OUT) :: IDX
    INTEGER(8), INTENT(OUT) :: H
    H = STATE_HASH(X)
    CALL PROBE(FSP, X, H, IDX)
  END SUBROUTINE LOOKUP

  ! the same with the hash already known
  SUBROUTINE PROBE(FSP, X, H, IDX)
    CLASS(FINITE_STATE_PROJECTION), INTENT(IN) :: FSP
    INTEGER, INTENT(IN) :: X(:)
    INTEGER(8), INTENT(IN) :: H
    INTEGER, INTENT(OUT) :: IDX
    INTEGER :: MASK, SLOT, J, K
    INTEGER(8) :: E, TAG
    LOGICAL :: SAME
    MASK = FSP%KTLEN - 1
    SLOT = INT(IAND(H, INT(MASK, 8))) + 1
    TAG = ISHFT(H, -8)
    DO
       E = FSP%KEYTAB(SLOT)
       IF (E == 0_8) THEN
          IDX = 0
          RETURN
       ENDIF
       IF (IAND(ISHFT(E, -32), LOW32) == TAG) THEN
          J = INT(IAND(E, LOW32))
          SAME = .TRUE.
          DO K = 1, SIZE(X)
             IF (FSP%STATE(K, J) /= X(K)) THEN
                SAME = .FALSE.
                EXIT
             ENDIF
          ENDDO
          IF (SAME) THEN
             IDX = J
             RETURN
          ENDIF
       ENDIF
       SLOT = IAND(SLOT, MASK) + 1
    ENDDO
  END SUBROUTINE PROBE

  ! enter (hash H -> index IDX) into the first free slot of its probe sequence
  SUBROUTINE TABLE_INSERT(FSP, H, IDX)
    CLASS(FINITE_STATE_PROJECTION), INTENT(INOUT) :: FSP
    INTEGER(8), INTENT(IN) :: H
    INTEGER, INTENT(IN) :: IDX
    INTEGER :: MASK, SLOT
    MASK = FSP%KTLEN - 1
    SLOT = INT(IAND(H, INT(MASK, 8))) + 1
    DO WHILE (FSP%KEYTAB(SLOT) /= 0_8)
       SLOT = IAND(SLOT, MASK) + 1
    ENDDO
    FSP%KEYTAB(SLOT) = IOR(ISHFT(ISHFT(H, -8), 32), INT(IDX, 8))
  END SUBROUTINE TABLE_INSERT

  ! enter the listed states ILO..IHI (none of them in the table yet) into the
  ! table.  Threaded: every thread owns a contiguous range of slots and enters the
  ! states whose probe sequence starts AND ends inside it (it reads all keys,
  ! writes only its own slots); the few sequences that would leave a range are
  ! entered afterwards.  Which slot a state gets differs from the one-by-one
  ! order, what LOOKUP returns does not.  CLEAR empties the table first.
  SUBROUTINE INSERT_RANGE(FSP, ILO, IHI, CLEAR)
    CLASS(FINITE_STATE_PROJECTION), INTENT(INOUT) :: FSP
    INTEGER, INTENT(IN) :: ILO, IHI
    LOGICAL, INTENT(IN) :: CLEAR
    INTEGER, PARAMETER :: AHEAD = 16, MAXLATE = 65536
    INTEGER :: I, MASK, NT, NTH, TID, SLOT, NLATE, P
    INTEGER(8) :: TOUCH, LO, HI
    INTEGER, ALLOCATABLE :: LATE(:)
    MASK = FSP%KTLEN - 1
    NT = HOST_THREADS(IHI - ILO + 1, 65536)
    IF (NT > 1) THEN
       ALLOCATE(LATE(MAXLATE))
       NLATE = 0
       NTH = 1
       !$OMP PARALLEL NUM_THREADS(NT) PRIVATE(TID, LO, HI, I, SLOT, P)
       TID = 0
       !$ TID = OMP_GET_THREAD_NUM()
       !$OMP SINGLE
       !$ NTH = OMP_GET_NUM_THREADS()
       !$OMP END SINGLE
       LO = 1 + INT(TID, 8) * FSP%KTLEN / NTH
       HI = INT(TID + 1, 8) * FSP%KTLEN / NTH
       IF (CLEAR) FSP%KEYTAB(LO:HI) = 0_8
       DO I = ILO, IHI
          SLOT = INT(IAND(FSP%KEY(I), INT(MASK, 8))) + 1
          IF (SLOT < LO .OR. SLOT > HI) CYCLE
          DO WHILE (SLOT <= HI)
             IF (FSP%KEYTAB(SLOT) == 0_8) EXIT
             SLOT = SLOT + 1
          ENDDO
          IF (SLOT <= HI) THEN
             FSP%KEYTAB(SLOT) = IOR(ISHFT(ISHFT(FSP%KEY(I), -8), 32), INT(I, 8))
          ELSE
             !$OMP ATOMIC CAPTURE
             NLATE = NLATE + 1
             P = NLATE
             !$OMP END ATOMIC
             IF (P <= MAXLATE) LATE(P) = I
          ENDIF
       ENDDO
       !$OMP END PARALLEL
       ! (more than MAXLATE leftovers cannot happen at load factor 1/2)
       IF (NLATE > MAXLATE) STOP 'KFSP STATESPACE: TABLE RANGE OVERFLOW'
       DO P = 1, NLATE
          CALL TABLE_INSERT(FSP, FSP%KEY(LATE(P)), LATE(P))
       ENDDO
       RETURN
    ENDIF
    IF (CLEAR) FSP%KEYTAB = 0_8
    TOUCH = 0
    DO I = ILO, IHI
       ! the slot of a later entry is requested now, so that it has arrived when
       ! its turn comes
       IF (I + AHEAD <= IHI) TOUCH = TOUCH + FSP%KEYTAB(INT(IAND(FSP%KEY(I + AHEAD), INT(MASK, 8))) + 1)
       CALL TABLE_INSERT(FSP, FSP%KEY(I), I)
    ENDDO
    TOUCH_SINK = INT(IAND(TOUCH, 1_8))
  END SUBROUTINE INSERT_RANGE

  ! The lists STATE / MATRIX%ADJ / OFFDIAG / DIAG (1:N) were written from outside (the solver's resident mode downloads
  ! them from the device when the solve is over): keys and look-up table are made to match.
  SUBROUTINE ADOPT_LISTS(FSP, MODEL, N)
    CLASS(FINITE_STATE_PROJECTION), INTENT(INOUT) :: FSP
    TYPE(CME_MODEL), INTENT(IN) :: MODEL
    INTEGER, INTENT(IN) :: N
    INTEGER :: I, NT
    FSP%SIZE = 0                                 ! (nothing to carry over into a regrown table)
    CALL RESERVE_TABLE(FSP, MAX(N, 1))
    NT = HOST_THREADS(N, 4096)
    !$OMP PARALLEL DO NUM_THREADS(NT) SCHEDULE(STATIC) IF(NT > 1)
    DO I = 1, N
       FSP%KEY(I) = STATE_HASH(FSP%STATE(1:MODEL%NSPECIES, I))
    ENDDO
    !$OMP END PARALLEL DO
    FSP%SIZE = N
    FSP%MATRIX%SIZE = N
    CALL REBUILD_TABLE(FSP)
  END SUBROUTINE ADOPT_LISTS

  SUBROUTINE REBUILD_TABLE(FSP)
    CLASS(FINITE_STATE_PROJECTION), INTENT(INOUT) :: FSP
    CALL INSERT_RANGE(FSP, 1, FSP%SIZE, .TRUE.)
  END SUBROUTINE REBUILD_TABLE

  ! make room for NEED listed states at load factor <= 1/2 (the table of the
  ! states listed so far is rebuilt when it has to grow)
  SUBROUTINE RESERVE_TABLE(FSP, NEED)
    CLASS(FINITE_STATE_PROJECTION), INTENT(INOUT) :: FSP
    INTEGER, INTENT(IN) :: NEED
    INTEGER :: P
    IF (2_8 * NEED <= FSP%KTLEN .AND. ALLOCATED(FSP%KEYTAB)) RETURN
    P = MAX(FSP%KTLEN, 1024)
    DO WHILE (P < 2_8 * NEED)
       P = 2 * P
    ENDDO
    ! load factor <= 1/4 right after growing keeps rebuilds rare
    IF (P < 4_8 * NEED) P = 2 * P
    IF (ALLOCATED(FSP%KEYTAB)) DEALLOCATE(FSP%KEYTAB)
    ALLOCATE(FSP%KEYTAB(P))
    FSP%KTLEN = P
    CALL REBUILD_TABLE(FSP)
  END SUBROUTINE RESERVE_TABLE

  DOUBLE PRECISION FUNCTION POINTWISE_FSP(FSP, X)
    CLASS(FINITE_STATE_PROJECTION), INTENT(IN) :: FSP
    INTEGER, INTENT(IN) :: X(:)
    INTEGER :: I
    I = INDEX_STATE(FSP, X)
    IF (I > 0) THEN
       POINTWISE_FSP = FSP%VECTOR(I)
    ELSE
       POINTWISE_FSP = 0.0D0
    ENDIF
  END FUNCTION POINTWISE_FSP

  INTEGER FUNCTION INDEX_STATE(FSP, X)
    CLASS(FINITE_STATE_PROJECTION), INTENT(IN) :: FSP
    INTEGER, INTENT(IN) :: X(:)
    INTEGER(8) :: H
    INDEX_STATE = 0
    IF (.NOT. LEGAL(X)) RETURN
    CALL LOOKUP(FSP, X, INDEX_STATE, H)
  END FUNCTION INDEX_STATE

  ! -------------------------------------------------------------- assembly

  ! put state Y (known to be absent, hash H) at the end of the list: table entry,
  ! zero probability, its propensities (StateSpace.f90:204-212) and an unlinked
  ! generator column
  SUBROUTINE APPEND_STATE(FSP, MODEL, Y, H)
    CLASS(FINITE_STATE_PROJECTION), INTENT(INOUT) :: FSP
    TYPE(CME_MODEL), INTENT(IN) :: MODEL
    INTEGER, INTENT(IN) :: Y(:)
    INTEGER(8), INTENT(IN) :: H
    INTEGER :: L, K
    DOUBLE PRECISION :: A
    L = FSP%SIZE + 1
    IF (2_8 * L > FSP%KTLEN) CALL RESERVE_TABLE(FSP, L)
    FSP%SIZE = L
    FSP%MATRIX%SIZE = L
    FSP%STATE(1:MODEL%NSPECIES, L) = Y(1:MODEL%NSPECIES)
    FSP%KEY(L) = H
    FSP%VECTOR(L) = 0.0D0
    CALL TABLE_INSERT(FSP, H, L)
    FSP%MATRIX%DIAG(L) = 0.0D0
    DO K = 1, MODEL%NREACTIONS
       A = MODEL%PROPENSITY(Y, K)
       FSP%MATRIX%DIAG(L) = FSP%MATRIX%DIAG(L) + A
       FSP%MATRIX%OFFDIAG(K, L) = A
       FSP%MATRIX%ADJ(K, L) = 0
    ENDDO
  END SUBROUTINE APPEND_STATE

  ! links of state number I (StateSpace.f90:213-244): its own column (successors
  ! present, -1 for a negative population) and, if BACK, the entries of the
  ! predecessors listed BEFORE ALO that now find it.  Reads the table only; the
  ! entries written belong to I alone (each (k, j) has a single successor).
  SUBROUTINE LINK_ONE(FSP, MODEL, I, ALO, BACK)
    CLASS(FINITE_STATE_PROJECTION), INTENT(INOUT) :: FSP
    TYPE(CME_MODEL), INTENT(IN) :: MODEL
    INTEGER, INTENT(IN) :: I, ALO
    LOGICAL, INTENT(IN) :: BACK
    INTEGER :: K, J, X(MODEL%NSPECIES), Y(MODEL%NSPECIES)
    INTEGER(8) :: H
    X = FSP%STATE(1:MODEL%NSPECIES, I)
    DO K = 1, MODEL%NREACTIONS
       Y = X + MODEL%STOICHIOMETRY(:, K)
       IF (ANY(Y < 0)) THEN
          FSP%MATRIX%ADJ(K, I) = -1
       ELSE
          J = 0
          IF (LEGAL(Y)) CALL LOOKUP(FSP, Y, J, H)
          FSP%MATRIX%ADJ(K, I) = J
       ENDIF
    ENDDO
    IF (.NOT. BACK) RETURN
    DO K = 1, MODEL%NREACTIONS
       Y = X - MODEL%STOICHIOMETRY(:, K)
       IF (.NOT. LEGAL(Y)) CYCLE
       CALL LOOKUP(FSP, Y, J, H)
       IF (J > 0 .AND. J < ALO) FSP%MATRIX%ADJ(K, J) = I
    ENDDO
  END SUBROUTINE LINK_ONE

  ! link the states ALO..AHI, all appended since the states before ALO were last
  ! fully linked
  SUBROUTINE LINK_NEW(FSP, MODEL, ALO, AHI, BACK)
    CLASS(FINITE_STATE_PROJECTION), INTENT(INOUT) :: FSP
    TYPE(CME_MODEL), INTENT(IN) :: MODEL
    INTEGER, INTENT(IN) :: ALO, AHI
    LOGICAL, INTENT(IN) :: BACK
    INTEGER :: I, NT
    NT = HOST_THREADS(AHI - ALO + 1, 1024)
    !$OMP PARALLEL DO NUM_THREADS(NT) SCHEDULE(STATIC) IF(NT > 1)
    DO I = ALO, AHI
       CALL LINK_ONE(FSP, MODEL, I, ALO, BACK)
    ENDDO
    !$OMP END PARALLEL DO
  END SUBROUTINE LINK_NEW

  ! append one state (if it is new) and connect it.  KEYIN is accepted for
  ! source compatibility (the reference passes a precomputed key) and ignored.
  SUBROUTINE ADD_STATE(FSP, MODEL, STATE, KEYIN)
    CLASS(FINITE_STATE_PROJECTION) :: FSP
    TYPE(CME_MODEL), INTENT(IN) :: MODEL
    INTEGER :: STATE(:)
    INTEGER(8), INTENT(IN), OPTIONAL :: KEYIN
    INTEGER :: IDX
    INTEGER(8) :: H
    IF (.NOT. LEGAL(STATE(1:MODEL%NSPECIES))) RETURN
    IF (FSP%SIZE >= FSP%MAX_SIZE) RETURN
    CALL LOOKUP(FSP, STATE(1:MODEL%NSPECIES), IDX, H)
    IF (IDX > 0) RETURN
    CALL APPEND_STATE(FSP, MODEL, STATE(1:MODEL%NSPECIES), H)
    CALL LINK_ONE(FSP, MODEL, FSP%SIZE, FSP%SIZE, .TRUE.)
  END SUBROUTINE ADD_STATE

  ! build table and generator for the seed list FSP%STATE(:,1:FSP%SIZE)
  SUBROUTINE MATRIX_STARTER(FSP, MODEL)
    TYPE(CME_MODEL), INTENT(IN) :: MODEL
    TYPE(FINITE_STATE_PROJECTION) :: FSP
    INTEGER :: I, K, IDX, N
    INTEGER(8) :: H
    DOUBLE PRECISION :: A
    N = FSP%SIZE
    FSP%SIZE = 0                       ! nothing to carry over into a regrown table
    CALL RESERVE_TABLE(FSP, MAX(N, 1))
    FSP%KEYTAB = 0_8
    FSP%SIZE = N
    FSP%MATRIX%SIZE = N
    DO I = 1, N
       CALL LOOKUP(FSP, FSP%STATE(1:MODEL%NSPECIES, I), IDX, H)
       FSP%KEY(I) = H
       IF (IDX == 0) CALL TABLE_INSERT(FSP, H, I)
       FSP%MATRIX%DIAG(I) = 0.0D0
       DO K = 1, MODEL%NREACTIONS
          A = MODEL%PROPENSITY(FSP%STATE(1:MODEL%NSPECIES, I), K)
          FSP%MATRIX%DIAG(I) = FSP%MATRIX%DIAG(I) + A
          FSP%MATRIX%OFFDIAG(K, I) = A
       ENDDO
    ENDDO
    CALL LINK_NEW(FSP, MODEL, 1, N, .FALSE.)
  END SUBROUTINE MATRIX_STARTER

  LOGICAL FUNCTION CUSTOMPROP_IS_PURE()
    CHARACTER(LEN=8) :: BUF
    INTEGER :: L, ST
    CALL GET_ENVIRONMENT_VARIABLE('KFSP_HOST_PARALLEL_PROPENSITY', BUF, L, ST)
    CUSTOMPROP_IS_PURE = .FALSE.
    IF (ST == 0 .AND. L > 0) CUSTOMPROP_IS_PURE = BUF(1:1) == '1'
  END FUNCTION CUSTOMPROP_IS_PURE

  ! Pass 2 of ONESTEP_EXTENDER for a long candidate list.  Candidate C is the open
  ! link (CJ(C), CK(C)) whose target, hash CH(C), is not listed.  The reference
  ! appends each target when its first candidate comes up; so the new states are
  ! the DISTINCT targets in order of first appearance.  Threads split the
  ! candidates by hash (equal targets meet in the same thread), find for each the
  ! first candidate with the same target (REP), a sweep over REP numbers the new
  ! states, and the rest - state, key, table entry, the links of all candidates -
  ! is filled in in parallel.  Propensity functions are called afterwards in the
  ! reference's order (state by state) from this thread, unless the model is a
  ! parsed one, whose evaluator is known to be re-entrant.
  SUBROUTINE APPEND_CANDIDATES(FSP, MODEL, TOTAL, CJ, CK, CH)
    TYPE(FINITE_STATE_PROJECTION) :: FSP
    TYPE(CME_MODEL), INTENT(IN) :: MODEL
    INTEGER, INTENT(IN) :: TOTAL, CJ(:), CK(:)
    INTEGER(8), INTENT(IN) :: CH(:)
    INTEGER, ALLOCATABLE :: REP(:), NEWOF(:), LT(:)
    INTEGER :: SD, PD, NT, NTH, TID, C, R, S, K, L, N0, NNEW, MINE, LTLEN, SLOT
    INTEGER :: NU(MODEL%NSPECIES, MODEL%NREACTIONS)
    LOGICAL :: SAME, PARPROP
    DOUBLE PRECISION :: A
    SD = MODEL%NSPECIES
    PD = MODEL%NREACTIONS
    N0 = FSP%SIZE
    NU = MODEL%STOICHIOMETRY(1:SD, 1:PD)
    NT = HOST_THREADS(TOTAL, 1)
    ALLOCATE(REP(TOTAL), NEWOF(TOTAL))
    NTH = 1
    !$OMP PARALLEL NUM_THREADS(NT) PRIVATE(TID, C, R, S, MINE, LTLEN, LT, SLOT, SAME)
    TID = 0
    !$ TID = OMP_GET_THREAD_NUM()
    !$OMP SINGLE
    !$ NTH = OMP_GET_NUM_THREADS()
    !$OMP END SINGLE
    MINE = 0
    DO C = 1, TOTAL
       IF (MOD(ISHFT(CH(C), -16), INT(NTH, 8)) == TID) MINE = MINE + 1
    ENDDO
    LTLEN = 64
    DO WHILE (LTLEN < 2 * MINE)
       LTLEN = 2 * LTLEN
    ENDDO
    ALLOCATE(LT(LTLEN))
    LT = 0
    DO C = 1, TOTAL
       IF (MOD(ISHFT(CH(C), -16), INT(NTH, 8)) /= TID) CYCLE
       SLOT = INT(IAND(ISHFT(CH(C), -4), INT(LTLEN - 1, 8))) + 1
       DO
          R = LT(SLOT)
          IF (R == 0) THEN
             LT(SLOT) = C
             REP(C) = C
             EXIT
          ENDIF
          IF (CH(R) == CH(C)) THEN
             SAME = .TRUE.
             DO S = 1, SD
                IF (FSP%STATE(S, CJ(R)) + NU(S, CK(R)) /= FSP%STATE(S, CJ(C)) + NU(S, CK(C))) SAME = .FALSE.
             ENDDO
             IF (SAME) THEN
                REP(C) = R
                EXIT
             ENDIF
          ENDIF
          SLOT = IAND(SLOT, LTLEN - 1) + 1
       ENDDO
    ENDDO
    DEALLOCATE(LT)
    !$OMP END PARALLEL

    NNEW = 0
    DO C = 1, TOTAL
       IF (REP(C) == C) THEN
          NNEW = NNEW + 1
          NEWOF(C) = N0 + NNEW
       ENDIF
    ENDDO
    IF (N0 + NNEW >= FSP%MAX_SIZE) STOP 'OVERFLOW ERROR: FSP SIZE EXCEEDS MEMORY LIMIT.'
    FSP%SIZE = N0                       ! (what a regrown table is rebuilt from)
    CALL RESERVE_TABLE(FSP, N0 + NNEW)

    ! a user's CUSTOMPROP is called from this thread only, unless
    ! KFSP_HOST_PARALLEL_PROPENSITY=1 declares it free of side effects
    PARPROP = .NOT. ASSOCIATED(MODEL%CUSTOMPROP) .OR. CUSTOMPROP_IS_PURE()
    !$OMP PARALLEL DO NUM_THREADS(NT) SCHEDULE(STATIC) PRIVATE(L, S, K, A)
    DO C = 1, TOTAL
       IF (REP(C) /= C) CYCLE
       L = NEWOF(C)
       DO S = 1, SD
          FSP%STATE(S, L) = FSP%STATE(S, CJ(C)) + NU(S, CK(C))
       ENDDO
       FSP%KEY(L) = CH(C)
       FSP%VECTOR(L) = 0.0D0
       FSP%MATRIX%ADJ(1:PD, L) = 0
       IF (PARPROP) THEN
          FSP%MATRIX%DIAG(L) = 0.0D0
          DO K = 1, PD
             A = MODEL%PROPENSITY(FSP%STATE(1:SD, L), K)
             FSP%MATRIX%DIAG(L) = FSP%MATRIX%DIAG(L) + A
             FSP%MATRIX%OFFDIAG(K, L) = A
          ENDDO
       ENDIF
    ENDDO
    !$OMP END PARALLEL DO
    IF (.NOT. PARPROP) THEN
       DO L = N0 + 1, N0 + NNEW
          FSP%MATRIX%DIAG(L) = 0.0D0
          DO K = 1, PD
             A = MODEL%PROPENSITY(FSP%STATE(1:SD, L), K)
             FSP%MATRIX%DIAG(L) = FSP%MATRIX%DIAG(L) + A
             FSP%MATRIX%OFFDIAG(K, L) = A
          ENDDO
       ENDDO
    ENDIF
    FSP%SIZE = N0 + NNEW
    FSP%MATRIX%SIZE = FSP%SIZE
    IF (NNEW > 0) CALL INSERT_RANGE(FSP, N0 + 1, N0 + NNEW, .FALSE.)
    !$OMP PARALLEL DO NUM_THREADS(NT) SCHEDULE(STATIC)
    DO C = 1, TOTAL
       FSP%MATRIX%ADJ(CK(C), CJ(C)) = NEWOF(REP(C))
    ENDDO
    !$OMP END PARALLEL DO
  END SUBROUTINE APPEND_CANDIDATES

  ! The sweep with its integer work on the device: new states (in the reference's order) and the
  ! complete link array come back in place; what stays here is what only the host can do - the
  ! propensities of the new states (the model's own code), their hashes and table entries.
  LOGICAL FUNCTION ONESTEP_ON_DEVICE(FSP, MODEL) RESULT(DONE)
    TYPE(FINITE_STATE_PROJECTION) :: FSP
    TYPE(CME_MODEL), INTENT(IN) :: MODEL
    CHARACTER(LEN=16) :: ENV
    INTEGER :: L, STAT, RC, N0, NNEW, SD, PD, I, K, NT
    INTEGER(8) :: TCLK
    DOUBLE PRECISION :: A
    LOGICAL :: PARPROP, COLUMNS
    DONE = .FALSE.
    IF (.NOT. ASSOCIATED(ONESTEP_DEVICE)) RETURN
    IF (ONESTEP_DEVICE_MIN < 0) THEN
       ONESTEP_DEVICE_MIN = 20000
       CALL GET_ENVIRONMENT_VARIABLE('KFSP_DEVICE_ONESTEP_MIN', ENV, L, STAT)
       IF (STAT == 0 .AND. L > 0) READ(ENV(1:L), *, IOSTAT=STAT) ONESTEP_DEVICE_MIN
       CALL GET_ENVIRONMENT_VARIABLE('KFSP_DEVICE_ONESTEP', ENV, L, STAT)
       IF (STAT == 0 .AND. L > 0) THEN
          IF (ENV(1:1) == '0') ONESTEP_DEVICE_MIN = HUGE(1)
       ENDIF
    ENDIF
    N0 = FSP%SIZE
    IF (N0 < ONESTEP_DEVICE_MIN) RETURN
    SD = MODEL%NSPECIES
    PD = MODEL%NREACTIONS
    IF (SIZE(FSP%STATE, 1) /= SD .OR. SIZE(FSP%MATRIX%ADJ, 1) /= PD) RETURN
    CALL TICK(0, TCLK)
    RC = ONESTEP_DEVICE(SD, PD, MODEL%STOICHIOMETRY(1:SD, 1:PD), N0, FSP%STATE, FSP%MATRIX%ADJ, MAXNUMBERMOLECULES, &
         FSP%MAX_SIZE - 1, NNEW, FSP%MATRIX%OFFDIAG, FSP%MATRIX%DIAG, COLUMNS)
    IF (RC == -11) STOP 'OVERFLOW ERROR: FSP SIZE EXCEEDS MEMORY LIMIT.'
    IF (RC /= 0) RETURN                      ! (-9: too many (state, reaction) pairs for 32-bit ordinals) the host sweep takes over
    CALL TICK(1, TCLK)
    DONE = .TRUE.
    IF (NNEW == N0) RETURN
    CALL RESERVE_TABLE(FSP, NNEW)
    IF (COLUMNS) THEN
       ! the device made the propensity columns as well: what is left is the host's own look-up structure
       NT = HOST_THREADS(NNEW - N0, 4096)
       !$OMP PARALLEL DO NUM_THREADS(NT) SCHEDULE(STATIC) IF(NT > 1)
       DO I = N0 + 1, NNEW
          FSP%KEY(I) = STATE_HASH(FSP%STATE(1:SD, I))
          FSP%VECTOR(I) = 0.0D0
       ENDDO
       !$OMP END PARALLEL DO
    ELSE
       PARPROP = .NOT. ASSOCIATED(MODEL%CUSTOMPROP) .OR. CUSTOMPROP_IS_PURE()
       NT = HOST_THREADS(NNEW - N0, 1024)
       !$OMP PARALLEL DO NUM_THREADS(NT) SCHEDULE(STATIC) IF(NT > 1) PRIVATE(K, A)
       DO I = N0 + 1, NNEW
          FSP%KEY(I) = STATE_HASH(FSP%STATE(1:SD, I))
          FSP%VECTOR(I) = 0.0D0
          IF (PARPROP) THEN
             FSP%MATRIX%DIAG(I) = 0.0D0
             DO K = 1, PD
                A = MODEL%PROPENSITY(FSP%STATE(1:SD, I), K)
                FSP%MATRIX%DIAG(I) = FSP%MATRIX%DIAG(I) + A
                FSP%MATRIX%OFFDIAG(K, I) = A
             ENDDO
          ENDIF
       ENDDO
       !$OMP END PARALLEL DO
       IF (.NOT. PARPROP) THEN
          DO I = N0 + 1, NNEW
             FSP%MATRIX%DIAG(I) = 0.0D0
             DO K = 1, PD
                A = MODEL%PROPENSITY(FSP%STATE(1:SD, I), K)
                FSP%MATRIX%DIAG(I) = FSP%MATRIX%DIAG(I) + A
                FSP%MATRIX%OFFDIAG(K, I) = A
             ENDDO
          ENDDO
       ENDIF
    ENDIF
    FSP%SIZE = NNEW
    FSP%MATRIX%SIZE = NNEW
    CALL INSERT_RANGE(FSP, N0 + 1, NNEW, .FALSE.)
    CALL TICK(2, TCLK)
  END FUNCTION ONESTEP_ON_DEVICE

  ! add every state one reaction away from the current list (in list order,
  ! reaction order; new states are appended and NOT revisited in this sweep).
  ! Pass 1 (parallel) resolves the open links whose target is already listed and
  ! collects the others per block of states; pass 2 (sequential, same (j, k)
  ! order as the reference's double loop) appends the targets still absent;
  ! pass 3 (parallel) links the appended states.
  SUBROUTINE ONESTEP_EXTENDER(FSP, MODEL)
    TYPE(FINITE_STATE_PROJECTION) :: FSP
    TYPE(CME_MODEL), INTENT(IN) :: MODEL
    INTEGER :: J, K, N0, IDX, SD, PD, NT, NTH, TID, T, C, Y(MODEL%NSPECIES)
    INTEGER(8) :: H, LO, HI, TOTAL, TCLK
    INTEGER(8), ALLOCATABLE :: OFFS(:)
    INTEGER, ALLOCATABLE :: NC(:), CJ(:), CK(:)
    INTEGER(8), ALLOCATABLE :: CH(:)
    INTEGER, PARAMETER :: AHEAD = 12
    INTEGER(8) :: TOUCH
    SD = MODEL%NSPECIES
    PD = MODEL%NREACTIONS
    N0 = FSP%SIZE
    CALL TICK(0, TCLK)
    IF (ONESTEP_ON_DEVICE(FSP, MODEL)) RETURN
    NT = HOST_THREADS(N0, 4096)
    ALLOCATE(OFFS(0:NT), NC(0:NT))
    OFFS = 0
    NC = 0
    NTH = 1
    !$OMP PARALLEL NUM_THREADS(NT) IF(NT > 1) PRIVATE(TID, LO, HI, J, K, C, Y, IDX, H)
    TID = 0
    !$ TID = OMP_GET_THREAD_NUM()
    !$OMP SINGLE
    !$ NTH = OMP_GET_NUM_THREADS()
    !$OMP END SINGLE
    LO = 1 + INT(TID, 8) * N0 / NTH
    HI = INT(TID + 1, 8) * N0 / NTH
    C = 0
    DO J = INT(LO), INT(HI)
       DO K = 1, PD
          IF (FSP%MATRIX%ADJ(K, J) == 0) C = C + 1
       ENDDO
    ENDDO
    OFFS(TID + 1) = C
    !$OMP BARRIER
    !$OMP SINGLE
    DO T = 1, NTH
       OFFS(T) = OFFS(T) + OFFS(T - 1)
    ENDDO
    TOTAL = OFFS(NTH)
    ALLOCATE(CJ(MAX(TOTAL, 1_8)), CK(MAX(TOTAL, 1_8)), CH(MAX(TOTAL, 1_8)))
    !$OMP END SINGLE
    C = 0
    DO J = INT(LO), INT(HI)
       DO K = 1, PD
          IF (FSP%MATRIX%ADJ(K, J) /= 0) CYCLE
          Y = FSP%STATE(1:SD, J) + MODEL%STOICHIOMETRY(:, K)
          IF (.NOT. LEGAL(Y)) CYCLE
          CALL LOOKUP(FSP, Y, IDX, H)
          IF (IDX > 0) THEN
             FSP%MATRIX%ADJ(K, J) = IDX
          ELSE
             C = C + 1
             CJ(OFFS(TID) + C) = J
             CK(OFFS(TID) + C) = K
             CH(OFFS(TID) + C) = H
          ENDIF
       ENDDO
    ENDDO
    NC(TID) = C
    !$OMP END PARALLEL
    CALL TICK(1, TCLK)

    ! the candidates of all blocks, in order, as one list
    TOTAL = 0
    DO T = 0, NTH - 1
       IF (OFFS(T) /= TOTAL) THEN
          CJ(TOTAL + 1:TOTAL + NC(T)) = CJ(OFFS(T) + 1:OFFS(T) + NC(T))
          CK(TOTAL + 1:TOTAL + NC(T)) = CK(OFFS(T) + 1:OFFS(T) + NC(T))
          CH(TOTAL + 1:TOTAL + NC(T)) = CH(OFFS(T) + 1:OFFS(T) + NC(T))
       ENDIF
       TOTAL = TOTAL + NC(T)
    ENDDO
    IF (HOST_THREADS(INT(TOTAL), 16384) > 1) THEN
       CALL APPEND_CANDIDATES(FSP, MODEL, INT(TOTAL), CJ, CK, CH)
    ELSE
       TOUCH = 0
       DO C = 1, INT(TOTAL)
          ! request the table slot of a later candidate now (it is a cache miss)
          IF (C + AHEAD <= TOTAL) TOUCH = TOUCH + FSP%KEYTAB(INT(IAND(CH(C + AHEAD), INT(FSP%KTLEN - 1, 8))) + 1)
          J = CJ(C)
          K = CK(C)
          Y = FSP%STATE(1:SD, J) + MODEL%STOICHIOMETRY(:, K)
          H = CH(C)
          CALL PROBE(FSP, Y, H, IDX)
          IF (IDX == 0) THEN
             IF (FSP%SIZE >= FSP%MAX_SIZE) STOP 'OVERFLOW ERROR: FSP SIZE EXCEEDS MEMORY LIMIT.'
             CALL APPEND_STATE(FSP, MODEL, Y, H)
             IDX = FSP%SIZE
             IF (FSP%SIZE >= FSP%MAX_SIZE) STOP 'OVERFLOW ERROR: FSP SIZE EXCEEDS MEMORY LIMIT.'
          ENDIF
          FSP%MATRIX%ADJ(K, J) = IDX
       ENDDO
       TOUCH_SINK = INT(IAND(TOUCH, 1_8))
    ENDIF
    CALL TICK(2, TCLK)
    IF (FSP%SIZE > N0) CALL LINK_NEW(FSP, MODEL, N0 + 1, FSP%SIZE, .FALSE.)
    CALL TICK(3, TCLK)
  END SUBROUTINE ONESTEP_EXTENDER

  ! largest power-of-ten threshold whose sub-threshold mass stays below DSUM
  ! (StateSpace.f90:398-429).  The reference sweeps W once per threshold; here the
  ! sums of the first NLEV thresholds are accumulated in one sweep, each of them
  ! over the same entries in the same order, hence to the same bits.
  SUBROUTINE FIND_DROPTOL(SD, LSIZE, W, DROPTOL, DSUM)
    INTEGER :: SD, LSIZE
    DOUBLE PRECISION :: W(:), DROPTOL, DSUM
    INTEGER, PARAMETER :: NLEV = 24
    DOUBLE PRECISION :: S, TOLS(NLEV), SL(NLEV), WI
    INTEGER :: I, T
    TOLS(1) = 1.0D-08
    DO T = 2, NLEV
       TOLS(T) = TOLS(T - 1) / 10.0D0
    ENDDO
    SL = 0.0D0
    DO I = 1, LSIZE
       WI = W(I)
       IF (WI < TOLS(1) .AND. WI > 0) THEN
          T = 1
          DO WHILE (WI < TOLS(T))
             SL(T) = SL(T) + WI
             T = T + 1
             IF (T > NLEV) EXIT
          ENDDO
       ENDIF
    ENDDO
    DO T = 1, NLEV
       IF (SL(T) < DSUM) THEN
          DROPTOL = TOLS(T)
          RETURN
       ENDIF
    ENDDO
    DROPTOL = TOLS(NLEV) / 10.0D0
    DO
       S = 0.0D0
       DO I = 1, LSIZE
          IF (W(I) < DROPTOL .AND. W(I) > 0) S = S + W(I)
       ENDDO
       IF (S < DSUM) EXIT
       DROPTOL = DROPTOL / 10.0D0
    ENDDO
  END SUBROUTINE FIND_DROPTOL

  ! The decision and compaction of DROP_STATES given AW = A*W:
  ! mark w < droptol, un-mark where (A w)_i > 1e-8, compact if more than 10 %
  ! are marked (StateSpace.f90:470-546).  CHANGED tells whether the FSP changed.
  SUBROUTINE DROP_STATES_CORE(W, FSP, MODEL, DSUM, AW, CHANGED)
    DOUBLE PRECISION :: W(:)
    TYPE(FINITE_STATE_PROJECTION) :: FSP
    TYPE(CME_MODEL), INTENT(IN) :: MODEL
    DOUBLE PRECISION :: DSUM
    DOUBLE PRECISION, INTENT(IN) :: AW(:)
    LOGICAL, INTENT(OUT) :: CHANGED
    LOGICAL, ALLOCATABLE :: DROP(:)
    LOGICAL :: MARK
    DOUBLE PRECISION :: DROPTOL
    INTEGER :: I, N, CNT, NT
    INTEGER(8) :: TCLK
    N = FSP%SIZE
    CHANGED = .FALSE.
    CALL TICK(0, TCLK)
    CALL FIND_DROPTOL(MODEL%NSPECIES, N, W, DROPTOL, DSUM)
    ALLOCATE(DROP(N))
    NT = HOST_THREADS(N, 65536)
    CNT = 0
    !$OMP PARALLEL DO NUM_THREADS(NT) SCHEDULE(STATIC) IF(NT > 1) REDUCTION(+:CNT) PRIVATE(MARK)
    DO I = 1, N
       MARK = W(I) < DROPTOL
       IF (MARK) CNT = CNT + 1
       IF (AW(I) > 1.0D-8) THEN
          MARK = .FALSE.
          CNT = CNT - 1          ! decremented whether or not it was marked (:490-495)
       ENDIF
       DROP(I) = MARK
    ENDDO
    !$OMP END PARALLEL DO
    CALL TICK(6, TCLK)
    IF (CNT * 1.0D0 / (N * 1.0D0) <= 0.1D0) RETURN
    CALL DROP_COMPACT(FSP, MODEL, DROP, W)
    CHANGED = .TRUE.
  END SUBROUTINE DROP_STATES_CORE

  ! The same compaction when the decision was taken elsewhere (on the device, kfsp_drop_plan):
  ! DROPPED(i) /= 0 marks the states to remove; the probability vector is compacted where it lives.
  SUBROUTINE DROP_APPLY_FLAGS(FSP, MODEL, DROPPED)
    TYPE(FINITE_STATE_PROJECTION) :: FSP
    TYPE(CME_MODEL), INTENT(IN) :: MODEL
    INTEGER(1), INTENT(IN) :: DROPPED(:)
    LOGICAL, ALLOCATABLE :: DROP(:)
    INTEGER(8) :: TCLK
    CALL TICK(0, TCLK)
    ALLOCATE(DROP(FSP%SIZE))
    DROP = DROPPED(1:FSP%SIZE) /= 0_1
    CALL TICK(6, TCLK)
    CALL DROP_COMPACT(FSP, MODEL, DROP)
  END SUBROUTINE DROP_APPLY_FLAGS

  ! remove the states marked in DROP from the list, the columns and (when given) W; list order is
  ! kept, links are renumbered, the table is rebuilt (StateSpace.f90:500-546)
  SUBROUTINE DROP_COMPACT(FSP, MODEL, DROP, W)
    TYPE(FINITE_STATE_PROJECTION) :: FSP
    TYPE(CME_MODEL), INTENT(IN) :: MODEL
    LOGICAL, INTENT(IN) :: DROP(:)
    DOUBLE PRECISION, OPTIONAL :: W(:)
    INTEGER, ALLOCATABLE :: NEWIDX(:)
    INTEGER :: I, J, K, Q, N, SD, PD, NT, JF, NMOVE
    INTEGER(8) :: TCLK
    DOUBLE PRECISION, ALLOCATABLE :: SCRD(:, :)
    INTEGER, ALLOCATABLE :: SCRI(:, :)
    INTEGER(8), ALLOCATABLE :: SCRK(:)
    LOGICAL :: HAVE_W
    HAVE_W = PRESENT(W)
    SD = MODEL%NSPECIES
    PD = MODEL%NREACTIONS
    N = FSP%SIZE
    CALL TICK(0, TCLK)
    ! new numbers of the states that stay (list order is kept, :500-546)
    ALLOCATE(NEWIDX(N))
    Q = 0
    JF = 0                             ! first state that moves
    DO J = 1, N
       IF (DROP(J)) THEN
          NEWIDX(J) = 0
          IF (JF == 0) JF = J
       ELSE
          Q = Q + 1
          NEWIDX(J) = Q
       ENDIF
    ENDDO
    NT = HOST_THREADS(N, 65536)
    ! every array is gathered into scratch storage and copied back, both in
    ! parallel: the sweep is bound by memory bandwidth, which one thread cannot use
    NMOVE = Q - JF + 1                 ! states JF..N that stay go to JF..Q
    IF (JF == 0) NMOVE = 0
    IF (NMOVE > 0) THEN
       ALLOCATE(SCRD(MAX(PD, 2), NMOVE), SCRI(MAX(PD, SD), NMOVE), SCRK(NMOVE))
       !$OMP PARALLEL NUM_THREADS(NT) IF(NT > 1) PRIVATE(J, I)
       !$OMP DO SCHEDULE(STATIC)
       DO J = JF, N
          I = NEWIDX(J) - JF + 1
          IF (I > 0) THEN
             SCRD(1:PD, I) = FSP%MATRIX%OFFDIAG(1:PD, J)
             SCRI(1:PD, I) = FSP%MATRIX%ADJ(1:PD, J)
             SCRK(I) = FSP%KEY(J)
          ENDIF
       ENDDO
       !$OMP END DO
       !$OMP DO SCHEDULE(STATIC)
       DO I = 1, NMOVE
          FSP%MATRIX%OFFDIAG(1:PD, JF + I - 1) = SCRD(1:PD, I)
          FSP%MATRIX%ADJ(1:PD, JF + I - 1) = SCRI(1:PD, I)
          FSP%KEY(JF + I - 1) = SCRK(I)
       ENDDO
       !$OMP END DO
       !$OMP DO SCHEDULE(STATIC)
       DO J = JF, N
          I = NEWIDX(J) - JF + 1
          IF (I > 0) THEN
             SCRD(1, I) = FSP%MATRIX%DIAG(J)
             IF (HAVE_W) SCRD(2, I) = W(J)
             SCRI(1:SD, I) = FSP%STATE(1:SD, J)
          ENDIF
       ENDDO
       !$OMP END DO
       !$OMP DO SCHEDULE(STATIC)
       DO I = 1, NMOVE
          FSP%MATRIX%DIAG(JF + I - 1) = SCRD(1, I)
          IF (HAVE_W) W(JF + I - 1) = SCRD(2, I)
          FSP%STATE(1:SD, JF + I - 1) = SCRI(1:SD, I)
       ENDDO
       !$OMP END DO
       !$OMP END PARALLEL
       DEALLOCATE(SCRD, SCRI, SCRK)
    ENDIF
    IF (HAVE_W) W(Q + 1:N) = 0.0D0
    FSP%SIZE = Q
    FSP%MATRIX%SIZE = Q
    CALL TICK(7, TCLK)
    !$OMP PARALLEL DO NUM_THREADS(NT) SCHEDULE(STATIC) IF(NT > 1) PRIVATE(K, I)
    DO J = 1, Q
       DO K = 1, PD
          I = FSP%MATRIX%ADJ(K, J)
          IF (I > 0) FSP%MATRIX%ADJ(K, J) = NEWIDX(I)
       ENDDO
    ENDDO
    !$OMP END PARALLEL DO
    CALL TICK(8, TCLK)
    CALL REBUILD_TABLE(FSP)
    CALL TICK(9, TCLK)
  END SUBROUTINE DROP_COMPACT

  ! reference signature: FMATVEC(X, Y, MATRIX) computes Y = A X
  SUBROUTINE DROP_STATES(W, FSP, MODEL, DSUM, FMATVEC)
    DOUBLE PRECISION :: W(:)
    TYPE(FINITE_STATE_PROJECTION) :: FSP
    DOUBLE PRECISION :: DSUM
    TYPE(CME_MODEL), INTENT(IN) :: MODEL
    EXTERNAL :: FMATVEC
    DOUBLE PRECISION, ALLOCATABLE :: AW(:)
    LOGICAL :: CHANGED
    ALLOCATE(AW(FSP%SIZE))
    CALL FMATVEC(W, AW, FSP%MATRIX)
    CALL DROP_STATES_CORE(W, FSP, MODEL, DSUM, AW, CHANGED)
  END SUBROUTINE DROP_STATES

  ! Is the processor's RANDOM_NUMBER the generator of LLVM flang's runtime - the
  ! minimal-standard congruential sequence s <- 48271 s mod (2**31 - 1) with
  ! RANDOM_SEED(GET/PUT) exchanging s itself, one double from two consecutive
  ! terms?  Decided by experiment at every call: the next four numbers and the
  ! seed after them must agree, and PUT must be read back by GET; the generator
  ! is left as it was found.  OK = .FALSE. leaves everything to RANDOM_NUMBER.
  SUBROUTINE LCG_RECOGNISED(OK, S)
    LOGICAL, INTENT(OUT) :: OK
    INTEGER(8), INTENT(OUT) :: S
    INTEGER :: NSEED, I
    INTEGER :: SEED(1), SEED1(1)
    INTEGER(8) :: G1, G2, Q
    DOUBLE PRECISION :: X(4), Y
    OK = .FALSE.
    S = 0
    CALL RANDOM_SEED(SIZE=NSEED)
    IF (NSEED /= 1) RETURN
    CALL RANDOM_SEED(GET=SEED)
    IF (SEED(1) < 1 .OR. SEED(1) >= LCG_M) RETURN
    CALL RANDOM_NUMBER(X)
    CALL RANDOM_SEED(GET=SEED1)
    Q = SEED(1)
    OK = .TRUE.
    DO I = 1, 4
       G1 = Q
       Q = MOD(Q * LCG_A, LCG_M)
       G2 = Q
       Q = MOD(Q * LCG_A, LCG_M)
       Y = DBLE(ISHFT(IOR(ISHFT(G1, 30), IAND(G2 - 1_8, LCG_LOW)), -7)) * LCG_SCALE
       IF (Y /= X(I)) OK = .FALSE.
    ENDDO
    IF (Q /= SEED1(1)) OK = .FALSE.
    CALL RANDOM_SEED(PUT=SEED)
    CALL RANDOM_SEED(GET=SEED1)
    IF (SEED1(1) /= SEED(1)) THEN
       ! PUT does not restore what GET gave: give the four numbers up for lost
       OK = .FALSE.
       RETURN
    ENDIF
    S = SEED(1)
  END SUBROUTINE LCG_RECOGNISED

  ! grow the FSP along one Gillespie path per listed state, each of duration
  ! TIMESTEP at most (StateSpace.f90:550-630); two uniform numbers per jump, in
  ! the reference's order, so that the same generator gives the same paths.
  ! The walk is sequential (the paths share one random stream and see the states
  ! added by earlier paths); a state met for the first time is appended with its
  ! propensities only, an open link (0) is resolved through the table as the
  ! reference does for its own open links, and the columns of all appended
  ! states are linked together afterwards.
  SUBROUTINE SSA_EXTENDER(TIMESTEP, FSP, MODEL)
    DOUBLE PRECISION :: TIMESTEP
    TYPE(FINITE_STATE_PROJECTION) :: FSP
    TYPE(CME_MODEL), INTENT(IN) :: MODEL
    INTEGER :: SD, PD, J, J0, K, S, N0, IDX, TOUCH, X(MODEL%NSPECIES), Y(MODEL%NSPECIES)
    INTEGER :: NU(MODEL%NSPECIES, MODEL%NREACTIONS)
    LOGICAL :: NEG
    INTEGER(8) :: H
    DOUBLE PRECISION :: T, R1, R2, R2A, ACC
    ! the uniform numbers are drawn NRB at a time (one runtime call instead of
    ! NRB); what is left of the last block is given back at the end, so the
    ! generator is left exactly where one call per number would leave it
    INTEGER, PARAMETER :: NRB = 2048
    DOUBLE PRECISION :: RB(NRB)
    INTEGER :: RP, NSEED
    INTEGER, ALLOCATABLE :: SEED0(:)
    INTEGER(8) :: TCLK
    ! Where RANDOM_NUMBER is recognised (LCG_RECOGNISED) its numbers are computed
    ! here instead, and the generator is set to where it would stand at the end.
    LOGICAL :: FAST
    INTEGER(8) :: RS, G1, G2
    IF (SSA_STREAMS_REQUESTED()) THEN
       CALL SSA_EXTENDER_STREAMS(TIMESTEP, FSP, MODEL)
       RETURN
    ENDIF
    SD = MODEL%NSPECIES
    PD = MODEL%NREACTIONS
    N0 = FSP%SIZE
    CALL RANDOM_SEED(SIZE=NSEED)
    ALLOCATE(SEED0(NSEED))
    RP = NRB
    CALL LCG_RECOGNISED(FAST, RS)
    NU = MODEL%STOICHIOMETRY(1:SD, 1:PD)
    TOUCH = 0
    CALL TICK(0, TCLK)
    PATHS: DO J0 = 1, N0
       J = J0
       X = FSP%STATE(1:SD, J)
       T = 0.0D0
       DO
          ! start fetching the links of J now: which one is needed is known only
          ! after the propensities have arrived
          TOUCH = TOUCH + FSP%MATRIX%ADJ(1, J) + FSP%MATRIX%ADJ(PD, J)
          IF (FAST) THEN
             G1 = RS
             RS = MOD(RS * LCG_A, LCG_M)
             G2 = RS
             RS = MOD(RS * LCG_A, LCG_M)
             R1 = DBLE(ISHFT(IOR(ISHFT(G1, 30), IAND(G2 - 1_8, LCG_LOW)), -7)) * LCG_SCALE
             G1 = RS
             RS = MOD(RS * LCG_A, LCG_M)
             G2 = RS
             RS = MOD(RS * LCG_A, LCG_M)
             R2 = DBLE(ISHFT(IOR(ISHFT(G1, 30), IAND(G2 - 1_8, LCG_LOW)), -7)) * LCG_SCALE
          ELSE
             IF (RP == NRB) THEN
                CALL RANDOM_SEED(GET=SEED0)
                CALL RANDOM_NUMBER(RB)
                RP = 0
             ENDIF
             R1 = RB(RP + 1)
             R2 = RB(RP + 2)
             RP = RP + 2
          ENDIF
          T = MIN(TIMESTEP, T + (-LOG(R1) / FSP%MATRIX%DIAG(J)))
          ! pick the reaction whose cumulative propensity first reaches r2*a0
          ACC = FSP%MATRIX%OFFDIAG(1, J)
          K = 1
          R2A = MIN(R2 * FSP%MATRIX%DIAG(J), FSP%MATRIX%DIAG(J))
          DO WHILE (ACC < R2A .AND. K < PD)
             K = K + 1
             ACC = ACC + FSP%MATRIX%OFFDIAG(K, J)
          ENDDO
          NEG = .FALSE.
          DO S = 1, SD
             Y(S) = X(S) + NU(S, K)
             NEG = NEG .OR. Y(S) < 0
          ENDDO
          IF (NEG) THEN
             FSP%MATRIX%ADJ(K, J) = -1
             EXIT
          ENDIF
          IF (FSP%MATRIX%ADJ(K, J) == 0) THEN
             IDX = 0
             IF (LEGAL(Y)) CALL LOOKUP(FSP, Y, IDX, H)
             IF (IDX > 0) THEN
                ! (a link the final linking would set to the same value: later
                ! paths through J need no look-up)
                FSP%MATRIX%ADJ(K, J) = IDX
                J = IDX
             ELSE
                IF (FSP%SIZE >= FSP%MAX_SIZE) EXIT PATHS
                IF (.NOT. LEGAL(Y)) EXIT
                CALL APPEND_STATE(FSP, MODEL, Y, H)
                FSP%MATRIX%ADJ(K, J) = FSP%SIZE
                J = FSP%SIZE
             ENDIF
          ELSE
             J = FSP%MATRIX%ADJ(K, J)
          ENDIF
          X = Y                      ! = FSP%STATE(1:SD, J)
          ! a path ends at the horizon or when it falls back onto an earlier seed
          IF (.NOT. (T < TIMESTEP .AND. J >= J0)) EXIT
       ENDDO
    ENDDO PATHS
    IF (FAST) THEN
       SEED0(1) = INT(RS)
       CALL RANDOM_SEED(PUT=SEED0)
    ELSE IF (RP < NRB) THEN
       CALL RANDOM_SEED(PUT=SEED0)
       IF (RP > 0) CALL RANDOM_NUMBER(RB(1:RP))
    ENDIF
    CALL TICK(4, TCLK)
    IF (FSP%SIZE > N0) CALL LINK_NEW(FSP, MODEL, N0 + 1, FSP%SIZE, .TRUE.)
    CALL TICK(5, TCLK)
    TOUCH_SINK = TOUCH
  END SUBROUTINE SSA_EXTENDER

  ! one path of SSA_EXTENDER_STREAMS: from listed state J0, on the stream of (SEEDMIX, J0);
  ! unlisted states it passes through are appended to the caller's record list
  SUBROUTINE STREAM_PATH(TIMESTEP, FSP, MODEL, NU, J0, SEEDMIX, RECX, RECJ, NREC, CAP)
    DOUBLE PRECISION, INTENT(IN) :: TIMESTEP
    TYPE(FINITE_STATE_PROJECTION), INTENT(IN) :: FSP
    TYPE(CME_MODEL), INTENT(IN) :: MODEL
    INTEGER, INTENT(IN) :: NU(:, :), J0
    INTEGER(8), INTENT(IN) :: SEEDMIX
    INTEGER, ALLOCATABLE, INTENT(INOUT) :: RECX(:, :), RECJ(:)
    INTEGER, INTENT(INOUT) :: NREC, CAP
    INTEGER :: SD, PD, J, K, S, IDX, X(MODEL%NSPECIES), Y(MODEL%NSPECIES)
    INTEGER, ALLOCATABLE :: TMPX(:, :), TMPJ(:)
    DOUBLE PRECISION :: TT, R1, R2, R2A, ACC, A0, PR(MODEL%NREACTIONS)
    INTEGER(8) :: H, RS, G1, G2
    LOGICAL :: NEG, VIRTUAL
    SD = MODEL%NSPECIES
    PD = MODEL%NREACTIONS
    ! the path's own stream: a 64-bit mix of (call, seed state) folded into the generator's range
    RS = IEOR(SEEDMIX * 2654435761_8, INT(J0, 8) * 40503_8 + 12345_8)
    RS = IAND(IEOR(RS, ISHFT(RS, -29)), LOW32) * 1181783497_8
    RS = 1_8 + MOD(IAND(IEOR(RS, ISHFT(RS, -32)), 9223372036854775807_8), LCG_M - 1_8)
    J = J0
    VIRTUAL = .FALSE.
    X = FSP%STATE(1:SD, J)
    TT = 0.0D0
    DO
       G1 = RS
       RS = MOD(RS * LCG_A, LCG_M)
       G2 = RS
       RS = MOD(RS * LCG_A, LCG_M)
       R1 = DBLE(ISHFT(IOR(ISHFT(G1, 30), IAND(G2 - 1_8, LCG_LOW)), -7)) * LCG_SCALE
       G1 = RS
       RS = MOD(RS * LCG_A, LCG_M)
       G2 = RS
       RS = MOD(RS * LCG_A, LCG_M)
       R2 = DBLE(ISHFT(IOR(ISHFT(G1, 30), IAND(G2 - 1_8, LCG_LOW)), -7)) * LCG_SCALE
       IF (R1 <= 0.0D0) R1 = LCG_SCALE
       IF (VIRTUAL) THEN
          A0 = 0.0D0
          DO K = 1, PD
             PR(K) = MODEL%PROPENSITY(X, K)
             A0 = A0 + PR(K)
          ENDDO
       ELSE
          A0 = FSP%MATRIX%DIAG(J)
          PR = FSP%MATRIX%OFFDIAG(1:PD, J)
       ENDIF
       IF (.NOT. (A0 > 0.0D0)) EXIT                 ! absorbing state
       TT = MIN(TIMESTEP, TT + (-KFSP_PLOG(R1) / A0))
       ACC = PR(1)
       K = 1
       R2A = MIN(R2 * A0, A0)
       DO WHILE (ACC < R2A .AND. K < PD)
          K = K + 1
          ACC = ACC + PR(K)
       ENDDO
       NEG = .FALSE.
       DO S = 1, SD
          Y(S) = X(S) + NU(S, K)
          NEG = NEG .OR. Y(S) < 0
       ENDDO
       IF (NEG) EXIT
       IDX = 0
       IF (.NOT. VIRTUAL) IDX = MAX(FSP%MATRIX%ADJ(K, J), 0)
       IF (IDX == 0) THEN
          IF (.NOT. LEGAL(Y)) EXIT
          CALL LOOKUP(FSP, Y, IDX, H)
       ENDIF
       X = Y
       IF (IDX > 0) THEN
          J = IDX
          VIRTUAL = .FALSE.
          IF (J < J0) EXIT                          ! fell back onto an earlier seed
       ELSE
          VIRTUAL = .TRUE.
          IF (NREC == CAP) THEN
             ALLOCATE(TMPX(SD, 2 * CAP), TMPJ(2 * CAP))
             TMPX(:, 1:CAP) = RECX
             TMPJ(1:CAP) = RECJ
             CALL MOVE_ALLOC(TMPX, RECX)
             CALL MOVE_ALLOC(TMPJ, RECJ)
             CAP = 2 * CAP
          ENDIF
          NREC = NREC + 1
          RECX(:, NREC) = Y
          RECJ(NREC) = J0
       ENDIF
       IF (.NOT. (TT < TIMESTEP)) EXIT
    ENDDO
  END SUBROUTINE STREAM_PATH

  ! log of 0 < X <= 1 by a FIXED sequence of IEEE operations - the waiting times of the independent-stream paths are
  ! defined through it rather than through the runtime's LOG, so that the device walk (plog of csrc/kfsp_ssa.hip, the
  ! same sequence) and this one produce the same bits.  X = M 2^E with M in [sqrt(1/2), sqrt(2)), S = (M - 1) / (M + 1),
  ! log M = 2 S (1 + S^2/3 + ... + S^22/23), log X = E ln2 + log M; about 1E-16 relative.
  DOUBLE PRECISION FUNCTION KFSP_PLOG(X) RESULT(R)
    DOUBLE PRECISION, INTENT(IN) :: X
    DOUBLE PRECISION :: M, F, S, Z, P, TWO_S, DE, HI, LO
    INTEGER :: E, K
    E = EXPONENT(X)
    M = FRACTION(X)                               ! X = M 2^E, M in [0.5, 1)
    IF (M < 0.70710678118654752440D0) THEN
       M = M + M
       E = E - 1
    ENDIF
    F = M - 1.0D0
    S = F / (2.0D0 + F)
    Z = S * S
    P = 1.0D0 / 23.0D0
    DO K = 21, 3, -2
       P = P * Z
       P = P + 1.0D0 / DBLE(K)
    ENDDO
    P = P * Z
    TWO_S = S + S
    R = TWO_S + TWO_S * P
    DE = DBLE(E)
    HI = DE * 6.93147180369123816490D-01
    LO = DE * 1.90821492927058770002D-10
    R = HI + (LO + R)
  END FUNCTION KFSP_PLOG

  ! KFSP_SSA_STREAMS=1: every path gets a random stream of its own (below)
  LOGICAL FUNCTION SSA_STREAMS_REQUESTED()
    CHARACTER(LEN=8) :: BUF
    INTEGER :: L, ST
    IF (SSA_STREAMS_FLAG < 0) THEN
       SSA_STREAMS_FLAG = 0
       CALL GET_ENVIRONMENT_VARIABLE('KFSP_SSA_STREAMS', BUF, L, ST)
       IF (ST == 0 .AND. L > 0) THEN
          IF (BUF(1:1) == '1') SSA_STREAMS_FLAG = 1
       ENDIF
    ENDIF
    SSA_STREAMS_REQUESTED = SSA_STREAMS_FLAG == 1
  END FUNCTION SSA_STREAMS_REQUESTED

  ! The walk of SSA_EXTENDER_STREAMS on the device: .TRUE. = the distinct unlisted states the paths met were appended
  ! (state, key, table entry, propensity columns; links are the caller's LINK_NEW), in the order the host's walk
  ! and append would have produced.
  LOGICAL FUNCTION SSA_STREAMS_ON_DEVICE(TIMESTEP, SEEDMIX, FSP, MODEL) RESULT(DONE)
    DOUBLE PRECISION, INTENT(IN) :: TIMESTEP
    INTEGER(8), INTENT(IN) :: SEEDMIX
    TYPE(FINITE_STATE_PROJECTION) :: FSP
    TYPE(CME_MODEL), INTENT(IN) :: MODEL
    CHARACTER(LEN=16) :: ENV
    INTEGER :: L, STAT, RC, N0, NF, SD, PD, I, NT
    DONE = .FALSE.
    IF (.NOT. ASSOCIATED(SSA_DEVICE)) RETURN
    IF (SSA_DEVICE_MIN < 0) THEN
       SSA_DEVICE_MIN = 20000
       CALL GET_ENVIRONMENT_VARIABLE('KFSP_DEVICE_SSA_MIN', ENV, L, STAT)
       IF (STAT == 0 .AND. L > 0) READ(ENV(1:L), *, IOSTAT=STAT) SSA_DEVICE_MIN
       CALL GET_ENVIRONMENT_VARIABLE('KFSP_DEVICE_SSA', ENV, L, STAT)
       IF (STAT == 0 .AND. L > 0) THEN
          IF (ENV(1:1) == '0') SSA_DEVICE_MIN = HUGE(1)
       ENDIF
    ENDIF
    N0 = FSP%SIZE
    IF (N0 < SSA_DEVICE_MIN) RETURN
    SD = MODEL%NSPECIES
    PD = MODEL%NREACTIONS
    IF (SIZE(FSP%STATE, 1) /= SD .OR. SIZE(FSP%MATRIX%ADJ, 1) /= PD) RETURN
    RC = SSA_DEVICE(TIMESTEP, SEEDMIX, SD, PD, MODEL%STOICHIOMETRY(1:SD, 1:PD), N0, FSP%STATE, FSP%MATRIX%ADJ, &
         FSP%MATRIX%OFFDIAG, FSP%MATRIX%DIAG, MAXNUMBERMOLECULES, FSP%MAX_SIZE - 1 - N0, NF)
    IF (RC /= 0) RETURN                       ! (no program, too many states for the list, ...: the host walks)
    DONE = .TRUE.
    IF (NF == 0) RETURN
    CALL RESERVE_TABLE(FSP, N0 + NF)
    NT = HOST_THREADS(NF, 4096)
    !$OMP PARALLEL DO NUM_THREADS(NT) SCHEDULE(STATIC) IF(NT > 1)
    DO I = N0 + 1, N0 + NF
       FSP%KEY(I) = STATE_HASH(FSP%STATE(1:SD, I))
       FSP%VECTOR(I) = 0.0D0
       FSP%MATRIX%ADJ(1:PD, I) = 0
    ENDDO
    !$OMP END PARALLEL DO
    FSP%SIZE = N0 + NF
    FSP%MATRIX%SIZE = N0 + NF
    CALL INSERT_RANGE(FSP, N0 + 1, N0 + NF, .FALSE.)
  END FUNCTION SSA_STREAMS_ON_DEVICE

  ! OPT-IN VARIANT, NOT the reference's sampling order (KFSP_SSA_STREAMS=1).
  ! The reference's paths share one random stream and each sees the states the
  ! earlier ones added, which makes the walk - two thirds of a large adaptive run -
  ! strictly sequential.  Here every path draws from a stream of its own (seeded
  ! from one number of RANDOM_NUMBER per call and the index of its seed state),
  ! walks the FSP as it stood when the call began, and continues through unlisted
  ! states by evaluating their propensities on the fly; the paths are independent
  ! and run on the thread team.  The states they met are then appended in path
  ! order (duplicates once) and linked.  The result is a valid expansion in the
  ! sense of the algorithm (any superset serves; the FSP criterion decides), it is
  ! the same for every number of threads, but states are found in another order
  ! than the reference finds them, so indices and later random numbers differ.
  ! A CUSTOMPROP function is called from the team only if declared free of side
  ! effects (KFSP_HOST_PARALLEL_PROPENSITY=1); otherwise one thread does all paths.
  SUBROUTINE SSA_EXTENDER_STREAMS(TIMESTEP, FSP, MODEL)
    DOUBLE PRECISION :: TIMESTEP
    TYPE(FINITE_STATE_PROJECTION) :: FSP
    TYPE(CME_MODEL), INTENT(IN) :: MODEL
    INTEGER :: SD, PD, N0, NT, NTH, TID, J0, IDX, NREC, CAP, T, I
    INTEGER :: NU(MODEL%NSPECIES, MODEL%NREACTIONS), Y(MODEL%NSPECIES)
    INTEGER, ALLOCATABLE :: RECX(:, :), RECJ(:)
    ! the records of all threads, concatenated after the walk
    INTEGER, ALLOCATABLE :: ALLX(:, :), ALLJ(:), CNT(:), OFS(:), HEAD(:), ORDER(:)
    DOUBLE PRECISION :: BASE
    INTEGER(8) :: H, SEEDMIX, TCLK
    SD = MODEL%NSPECIES
    PD = MODEL%NREACTIONS
    N0 = FSP%SIZE
    NU = MODEL%STOICHIOMETRY(1:SD, 1:PD)
    CALL TICK(0, TCLK)
    CALL RANDOM_NUMBER(BASE)
    SEEDMIX = INT(BASE * 2147483647.0D0, 8)
    IF (SSA_STREAMS_ON_DEVICE(TIMESTEP, SEEDMIX, FSP, MODEL)) THEN
       CALL TICK(4, TCLK)
       IF (FSP%SIZE > N0) CALL LINK_NEW(FSP, MODEL, N0 + 1, FSP%SIZE, .TRUE.)
       CALL TICK(5, TCLK)
       RETURN
    ENDIF
    NT = HOST_THREADS(N0, 1024)
    IF (ASSOCIATED(MODEL%CUSTOMPROP) .AND. .NOT. CUSTOMPROP_IS_PURE()) NT = 1
    ALLOCATE(CNT(0:NT), OFS(0:NT))
    CNT = 0
    NTH = 1
    !$OMP PARALLEL NUM_THREADS(NT) IF(NT > 1) PRIVATE(TID, J0, NREC, CAP, RECX, RECJ)
    TID = 0
    !$ TID = OMP_GET_THREAD_NUM()
    !$OMP SINGLE
    !$ NTH = OMP_GET_NUM_THREADS()
    !$OMP END SINGLE
    CAP = 1024
    ALLOCATE(RECX(SD, CAP), RECJ(CAP))
    NREC = 0
    !$OMP DO SCHEDULE(DYNAMIC, 256)
    DO J0 = 1, N0
       CALL STREAM_PATH(TIMESTEP, FSP, MODEL, NU, J0, SEEDMIX, RECX, RECJ, NREC, CAP)
    ENDDO
    !$OMP END DO
    CNT(TID + 1) = NREC
    !$OMP BARRIER
    !$OMP SINGLE
    OFS(0) = 0
    DO T = 1, NTH
       OFS(T) = OFS(T - 1) + CNT(T)
    ENDDO
    ALLOCATE(ALLX(SD, MAX(OFS(NTH), 1)), ALLJ(MAX(OFS(NTH), 1)))
    !$OMP END SINGLE
    IF (NREC > 0) THEN
       ALLX(:, OFS(TID) + 1:OFS(TID) + NREC) = RECX(:, 1:NREC)
       ALLJ(OFS(TID) + 1:OFS(TID) + NREC) = RECJ(1:NREC)
    ENDIF
    DEALLOCATE(RECX, RECJ)
    !$OMP END PARALLEL
    CALL TICK(4, TCLK)

    ! append in path order (seed state, then order along the path), first occurrence of a
    ! state wins.  A path's records sit together in one thread's list, but which thread
    ! walked which path depends on the schedule: a counting sort by seed state restores an
    ! order that does not.
    ALLOCATE(HEAD(N0 + 1))
    HEAD = 0
    DO I = 1, OFS(NTH)
       HEAD(ALLJ(I) + 1) = HEAD(ALLJ(I) + 1) + 1
    ENDDO
    HEAD(1) = 1
    DO J0 = 2, N0 + 1
       HEAD(J0) = HEAD(J0) + HEAD(J0 - 1)          ! HEAD(j) = first position of path j's records
    ENDDO
    ALLOCATE(ORDER(MAX(OFS(NTH), 1)))
    DO I = 1, OFS(NTH)
       ORDER(HEAD(ALLJ(I))) = I
       HEAD(ALLJ(I)) = HEAD(ALLJ(I)) + 1
    ENDDO
    DO I = 1, OFS(NTH)
       Y = ALLX(:, ORDER(I))
       IF (FSP%SIZE >= FSP%MAX_SIZE) EXIT
       CALL LOOKUP(FSP, Y, IDX, H)
       IF (IDX == 0) CALL APPEND_STATE(FSP, MODEL, Y, H)
    ENDDO
    IF (FSP%SIZE > N0) CALL LINK_NEW(FSP, MODEL, N0 + 1, FSP%SIZE, .TRUE.)
    CALL TICK(5, TCLK)
  END SUBROUTINE SSA_EXTENDER_STREAMS

  ! COMPUTE_RKEY (StateSpace.f90:635-669): the change of the reference's POSITIONAL state key,
  ! key(x) = 2 + sum_k x_k (MAXNUMBERMOLECULES + 1)**(k-1) (HashTable.f90:39-59), under each
  ! reaction: key(x + nu_j) = key(x) + RKEYSIGN(j) * REACTIONKEY(j).  Same arguments and the same
  ! integers as the reference returns; its 35-digit BIG_INTEGER is INTEGER(16) here (10001**9 <
  ! 2**127 covers nine species; more raise an error).  This module does not use positional keys
  ! itself (FSP%KEY holds 64-bit hashes, the table compares states), the routine is kept for
  ! drivers that call it.
  SUBROUTINE COMPUTE_RKEY(REACTIONKEY, RKEYSIGN, SD, PD, MODEL)
    INTEGER :: SD, PD
    INTEGER(16) :: REACTIONKEY(PD)
    INTEGER :: RKEYSIGN(PD)
    TYPE(CME_MODEL), INTENT(IN) :: MODEL
    INTEGER :: I, J, RS(SD), SGN
    INTEGER(16) :: RKEY, BASE, POW
    IF (SD > 9) STOP 'COMPUTE_RKEY: MORE THAN 9 SPECIES DO NOT FIT THE 128-BIT POSITIONAL KEY'
    BASE = INT(MAXNUMBERMOLECULES, 16) + 1_16
    DO J = 1, PD
       RS = MODEL%STOICHIOMETRY(1:SD, J)
       SGN = 1
       RKEY = 0_16
       POW = 1_16
       DO I = 1, SD
          IF (SGN * RS(I) < 0) THEN
             SGN = -SGN
             RKEY = INT(ABS(RS(I)), 16) * POW - RKEY
          ELSE
             RKEY = INT(ABS(RS(I)), 16) * POW + RKEY
          ENDIF
          POW = POW * BASE
       ENDDO
       REACTIONKEY(J) = RKEY
       RKEYSIGN(J) = SGN
    ENDDO
  END SUBROUTINE COMPUTE_RKEY

END MODULE STATESPACE
