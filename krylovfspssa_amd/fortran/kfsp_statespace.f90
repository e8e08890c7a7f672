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
MODULE STATESPACE
  USE MODELMODULE
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
     ! open-addressing table: KEYTAB = hash of the entry, KVTAB = state index (0 = free)
     INTEGER(8), ALLOCATABLE :: KEYTAB(:)
     INTEGER, ALLOCATABLE :: KVTAB(:)
   CONTAINS
     PROCEDURE :: CREATE => CREATE_FSP
     PROCEDURE :: CLEAR => CLEAR_FSP
     PROCEDURE :: PROBABILITY => POINTWISE_FSP
     PROCEDURE :: ADD => ADD_STATE
     PROCEDURE :: INDEX => INDEX_STATE
  END TYPE FINITE_STATE_PROJECTION

  PRIVATE :: STATE_HASH, LOOKUP, TABLE_INSERT, REBUILD_TABLE, LEGAL

CONTAINS

  SUBROUTINE CREATE_FSP(FSP, MODEL, MAX_SIZE_CUSTOM)
    CLASS(FINITE_STATE_PROJECTION), INTENT(INOUT) :: FSP
    TYPE(CME_MODEL), INTENT(IN) :: MODEL
    INTEGER, OPTIONAL, INTENT(IN) :: MAX_SIZE_CUSTOM
    INTEGER :: N, M, P
    N = MODEL%NSPECIES
    M = MODEL%NREACTIONS
    IF (PRESENT(MAX_SIZE_CUSTOM)) FSP%MAX_SIZE = MAX_SIZE_CUSTOM
    IF (ALLOCATED(FSP%STATE)) CALL CLEAR_FSP(FSP)
    ! table length: power of two, load factor <= 1/2
    P = 16
    DO WHILE (P < 2 * FSP%MAX_SIZE)
       P = 2 * P
    ENDDO
    FSP%KTLEN = P
    ALLOCATE(FSP%STATE(N, FSP%MAX_SIZE), FSP%KEY(FSP%MAX_SIZE), &
         FSP%MATRIX%DIAG(FSP%MAX_SIZE), FSP%MATRIX%OFFDIAG(M, FSP%MAX_SIZE), FSP%MATRIX%ADJ(M, FSP%MAX_SIZE), &
         FSP%KEYTAB(P), FSP%KVTAB(P), FSP%VECTOR(FSP%MAX_SIZE))
    FSP%SIZE = 0
    FSP%MATRIX%SIZE = 0
    FSP%KVTAB = 0
  END SUBROUTINE CREATE_FSP

  SUBROUTINE CLEAR_FSP(FSP)
    CLASS(FINITE_STATE_PROJECTION), INTENT(INOUT) :: FSP
    IF (ALLOCATED(FSP%STATE)) DEALLOCATE(FSP%STATE, FSP%KEY, FSP%MATRIX%DIAG, FSP%MATRIX%OFFDIAG, &
         FSP%MATRIX%ADJ, FSP%KEYTAB, FSP%KVTAB, FSP%VECTOR)
    FSP%SIZE = 0
    FSP%MATRIX%SIZE = 0
  END SUBROUTINE CLEAR_FSP

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

  ! index of state X in the FSP (0 = absent); H = its hash, SLOT = where it is
  ! or where it would be inserted
  SUBROUTINE LOOKUP(FSP, X, IDX, H, SLOT)
    CLASS(FINITE_STATE_PROJECTION), INTENT(IN) :: FSP
    INTEGER, INTENT(IN) :: X(:)
    INTEGER, INTENT(OUT) :: IDX, SLOT
    INTEGER(8), INTENT(OUT) :: H
    INTEGER :: MASK, J, K
    LOGICAL :: SAME
    H = STATE_HASH(X)
    MASK = FSP%KTLEN - 1
    SLOT = INT(IAND(H, INT(MASK, 8))) + 1
    DO
       J = FSP%KVTAB(SLOT)
       IF (J == 0) THEN
          IDX = 0
          RETURN
       ENDIF
       IF (FSP%KEYTAB(SLOT) == H) THEN
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
  END SUBROUTINE LOOKUP

  SUBROUTINE TABLE_INSERT(FSP, H, SLOT, IDX)
    CLASS(FINITE_STATE_PROJECTION), INTENT(INOUT) :: FSP
    INTEGER(8), INTENT(IN) :: H
    INTEGER, INTENT(IN) :: SLOT, IDX
    FSP%KEYTAB(SLOT) = H
    FSP%KVTAB(SLOT) = IDX
  END SUBROUTINE TABLE_INSERT

  SUBROUTINE REBUILD_TABLE(FSP)
    CLASS(FINITE_STATE_PROJECTION), INTENT(INOUT) :: FSP
    INTEGER :: I, SLOT, MASK
    FSP%KVTAB = 0
    MASK = FSP%KTLEN - 1
    DO I = 1, FSP%SIZE
       SLOT = INT(IAND(FSP%KEY(I), INT(MASK, 8))) + 1
       DO WHILE (FSP%KVTAB(SLOT) /= 0)
          SLOT = IAND(SLOT, MASK) + 1
       ENDDO
       CALL TABLE_INSERT(FSP, FSP%KEY(I), SLOT, I)
    ENDDO
  END SUBROUTINE REBUILD_TABLE

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
    INTEGER :: SLOT
    INTEGER(8) :: H
    INDEX_STATE = 0
    IF (.NOT. LEGAL(X)) RETURN
    CALL LOOKUP(FSP, X, INDEX_STATE, H, SLOT)
  END FUNCTION INDEX_STATE

  ! -------------------------------------------------------------- assembly

  ! column of the generator for state number I (already in the list and in the
  ! table): propensities, forward links to successors already present, and the
  ! back links of predecessors already present (StateSpace.f90:204-244)
  SUBROUTINE LINK_STATE(FSP, MODEL, I)
    CLASS(FINITE_STATE_PROJECTION), INTENT(INOUT) :: FSP
    TYPE(CME_MODEL), INTENT(IN) :: MODEL
    INTEGER, INTENT(IN) :: I
    INTEGER :: K, J, SLOT, X(MODEL%NSPECIES), Y(MODEL%NSPECIES)
    INTEGER(8) :: H
    DOUBLE PRECISION :: A
    X = FSP%STATE(1:MODEL%NSPECIES, I)
    FSP%MATRIX%DIAG(I) = 0.0D0
    DO K = 1, MODEL%NREACTIONS
       A = MODEL%PROPENSITY(X, K)
       FSP%MATRIX%DIAG(I) = FSP%MATRIX%DIAG(I) + A
       FSP%MATRIX%OFFDIAG(K, I) = A
       Y = X + MODEL%STOICHIOMETRY(:, K)
       IF (ANY(Y < 0)) THEN
          FSP%MATRIX%ADJ(K, I) = -1
       ELSE
          J = 0
          IF (LEGAL(Y)) CALL LOOKUP(FSP, Y, J, H, SLOT)
          FSP%MATRIX%ADJ(K, I) = J
       ENDIF
    ENDDO
    DO K = 1, MODEL%NREACTIONS
       Y = X - MODEL%STOICHIOMETRY(:, K)
       IF (.NOT. LEGAL(Y)) CYCLE
       CALL LOOKUP(FSP, Y, J, H, SLOT)
       IF (J > 0) FSP%MATRIX%ADJ(K, J) = I
    ENDDO
  END SUBROUTINE LINK_STATE

  ! append one state (if it is new) and connect it.  KEYIN is accepted for
  ! source compatibility (the reference passes a precomputed key) and ignored.
  SUBROUTINE ADD_STATE(FSP, MODEL, STATE, KEYIN)
    CLASS(FINITE_STATE_PROJECTION) :: FSP
    TYPE(CME_MODEL), INTENT(IN) :: MODEL
    INTEGER :: STATE(:)
    INTEGER(8), INTENT(IN), OPTIONAL :: KEYIN
    INTEGER :: IDX, SLOT, L
    INTEGER(8) :: H
    IF (.NOT. LEGAL(STATE(1:MODEL%NSPECIES))) RETURN
    IF (FSP%SIZE >= FSP%MAX_SIZE) RETURN
    CALL LOOKUP(FSP, STATE(1:MODEL%NSPECIES), IDX, H, SLOT)
    IF (IDX > 0) RETURN
    FSP%SIZE = FSP%SIZE + 1
    L = FSP%SIZE
    FSP%STATE(1:MODEL%NSPECIES, L) = STATE(1:MODEL%NSPECIES)
    FSP%KEY(L) = H
    FSP%VECTOR(L) = 0.0D0
    FSP%MATRIX%SIZE = L
    CALL TABLE_INSERT(FSP, H, SLOT, L)
    CALL LINK_STATE(FSP, MODEL, L)
  END SUBROUTINE ADD_STATE

  ! build table and generator for the seed list FSP%STATE(:,1:FSP%SIZE)
  SUBROUTINE MATRIX_STARTER(FSP, MODEL)
    TYPE(CME_MODEL), INTENT(IN) :: MODEL
    TYPE(FINITE_STATE_PROJECTION) :: FSP
    INTEGER :: I, IDX, SLOT
    INTEGER(8) :: H
    FSP%KVTAB = 0
    FSP%MATRIX%SIZE = FSP%SIZE
    DO I = 1, FSP%SIZE
       CALL LOOKUP(FSP, FSP%STATE(1:MODEL%NSPECIES, I), IDX, H, SLOT)
       FSP%KEY(I) = H
       IF (IDX == 0) CALL TABLE_INSERT(FSP, H, SLOT, I)
       CALL LINK_STATE(FSP, MODEL, I)
    ENDDO
  END SUBROUTINE MATRIX_STARTER

  ! add every state one reaction away from the current list (in list order,
  ! reaction order; new states are appended and NOT revisited in this sweep)
  SUBROUTINE ONESTEP_EXTENDER(FSP, MODEL)
    TYPE(FINITE_STATE_PROJECTION) :: FSP
    TYPE(CME_MODEL), INTENT(IN) :: MODEL
    INTEGER :: J, K, N0, IDX, SLOT, Y(MODEL%NSPECIES)
    INTEGER(8) :: H
    N0 = FSP%SIZE
    DO J = 1, N0
       DO K = 1, MODEL%NREACTIONS
          IF (FSP%MATRIX%ADJ(K, J) /= 0) CYCLE
          Y = FSP%STATE(1:MODEL%NSPECIES, J) + MODEL%STOICHIOMETRY(:, K)
          IF (.NOT. LEGAL(Y)) CYCLE
          CALL LOOKUP(FSP, Y, IDX, H, SLOT)
          IF (IDX > 0) THEN
             FSP%MATRIX%ADJ(K, J) = IDX
          ELSE
             CALL ADD_STATE(FSP, MODEL, Y)
             IF (FSP%SIZE >= FSP%MAX_SIZE) STOP 'OVERFLOW ERROR: FSP SIZE EXCEEDS MEMORY LIMIT.'
          ENDIF
       ENDDO
    ENDDO
  END SUBROUTINE ONESTEP_EXTENDER

  ! largest power-of-ten threshold whose sub-threshold mass stays below DSUM
  SUBROUTINE FIND_DROPTOL(SD, LSIZE, W, DROPTOL, DSUM)
    INTEGER :: SD, LSIZE
    DOUBLE PRECISION :: W(:), DROPTOL, DSUM
    DOUBLE PRECISION :: S
    INTEGER :: I
    DROPTOL = 1.0D-08
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
    INTEGER, ALLOCATABLE :: NEWIDX(:)
    DOUBLE PRECISION :: DROPTOL
    INTEGER :: I, J, K, Q, N, CNT, SD, PD
    SD = MODEL%NSPECIES
    PD = MODEL%NREACTIONS
    N = FSP%SIZE
    CHANGED = .FALSE.
    CALL FIND_DROPTOL(SD, N, W, DROPTOL, DSUM)
    ALLOCATE(DROP(N))
    CNT = 0
    DO I = 1, N
       DROP(I) = W(I) < DROPTOL
       IF (DROP(I)) CNT = CNT + 1
    ENDDO
    DO I = 1, N
       IF (AW(I) > 1.0D-8) THEN
          DROP(I) = .FALSE.
          CNT = CNT - 1          ! decremented whether or not it was marked (:490-495)
       ENDIF
    ENDDO
    IF (CNT * 1.0D0 / (N * 1.0D0) <= 0.1D0) RETURN

    ALLOCATE(NEWIDX(N))
    Q = 0
    DO J = 1, N
       IF (DROP(J)) THEN
          NEWIDX(J) = 0
       ELSE
          Q = Q + 1
          NEWIDX(J) = Q
          IF (Q /= J) THEN
             W(Q) = W(J)
             FSP%STATE(1:SD, Q) = FSP%STATE(1:SD, J)
             FSP%KEY(Q) = FSP%KEY(J)
             FSP%MATRIX%DIAG(Q) = FSP%MATRIX%DIAG(J)
             FSP%MATRIX%OFFDIAG(1:PD, Q) = FSP%MATRIX%OFFDIAG(1:PD, J)
             FSP%MATRIX%ADJ(1:PD, Q) = FSP%MATRIX%ADJ(1:PD, J)
          ENDIF
       ENDIF
    ENDDO
    W(Q + 1:N) = 0.0D0
    FSP%SIZE = Q
    FSP%MATRIX%SIZE = Q
    DO J = 1, Q
       DO K = 1, PD
          I = FSP%MATRIX%ADJ(K, J)
          IF (I > 0) FSP%MATRIX%ADJ(K, J) = NEWIDX(I)
       ENDDO
    ENDDO
    CALL REBUILD_TABLE(FSP)
    CHANGED = .TRUE.
  END SUBROUTINE DROP_STATES_CORE

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

  ! grow the FSP along one Gillespie path per listed state, each of duration
  ! TIMESTEP at most (StateSpace.f90:550-630); two uniform numbers per jump, in
  ! the reference's order, so that the same generator gives the same paths
  SUBROUTINE SSA_EXTENDER(TIMESTEP, FSP, MODEL)
    DOUBLE PRECISION :: TIMESTEP
    TYPE(FINITE_STATE_PROJECTION) :: FSP
    TYPE(CME_MODEL), INTENT(IN) :: MODEL
    INTEGER :: SD, PD, J, J0, K, N0, IDX, SLOT, X(MODEL%NSPECIES), Y(MODEL%NSPECIES)
    INTEGER(8) :: H
    DOUBLE PRECISION :: T, R1, R2, R2A, ACC
    SD = MODEL%NSPECIES
    PD = MODEL%NREACTIONS
    N0 = FSP%SIZE
    DO J0 = 1, N0
       J = J0
       X = FSP%STATE(1:SD, J)
       T = 0.0D0
       DO
          CALL RANDOM_NUMBER(R1)
          CALL RANDOM_NUMBER(R2)
          T = MIN(TIMESTEP, T + (-LOG(R1) / FSP%MATRIX%DIAG(J)))
          ! pick the reaction whose cumulative propensity first reaches r2*a0
          ACC = FSP%MATRIX%OFFDIAG(1, J)
          K = 1
          R2A = MIN(R2 * FSP%MATRIX%DIAG(J), FSP%MATRIX%DIAG(J))
          DO WHILE (ACC < R2A .AND. K < PD)
             K = K + 1
             ACC = ACC + FSP%MATRIX%OFFDIAG(K, J)
          ENDDO
          Y = X + MODEL%STOICHIOMETRY(:, K)
          IF (ANY(Y < 0)) THEN
             FSP%MATRIX%ADJ(K, J) = -1
             EXIT
          ENDIF
          IF (FSP%MATRIX%ADJ(K, J) == 0) THEN
             IDX = 0
             IF (LEGAL(Y)) CALL LOOKUP(FSP, Y, IDX, H, SLOT)
             IF (IDX > 0) THEN
                J = IDX
             ELSE
                IF (FSP%SIZE >= FSP%MAX_SIZE) RETURN
                IF (.NOT. LEGAL(Y)) EXIT
                CALL ADD_STATE(FSP, MODEL, Y)
                J = FSP%SIZE
             ENDIF
          ELSE
             J = FSP%MATRIX%ADJ(K, J)
          ENDIF
          X = FSP%STATE(1:SD, J)
          ! a path ends at the horizon or when it falls back onto an earlier seed
          IF (.NOT. (T < TIMESTEP .AND. J >= J0)) EXIT
       ENDDO
    ENDDO
  END SUBROUTINE SSA_EXTENDER

END MODULE STATESPACE
