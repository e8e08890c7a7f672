! MODULE MODELMODULE - the stochastic reaction network: stoichiometry,
! propensities (compiled-in through CUSTOMPROP or parsed from a models/*.input
! file) and parameter values.
!
! Host-side drop-in surface: same module name, type name, public components and
! type-bound procedures as the reference (src/model/ModelModule.f90:6-42), so
! that its drivers (examples/*.f90, test/*.f90) compile unchanged.  New code;
! deliberate differences, all towards accepting what the reference's own files
! contain:
!   * section keywords of the .input format are matched case-insensitively (the
!     shipped models use lower case, which the reference loader silently
!     ignores: ModelModule.f90:95-140 vs models/toggle_model.input:1-28)
!   * reaction terms are tokenised properly: optional integer coefficient
!     followed by an exact species name (ModelModule.f90:269-293 relies on a
!     suffix search and on SAVEd locals)
!   * nothing is echoed while parsing
MODULE MODELMODULE
  USE KFSP_EXPR
  IMPLICIT NONE

  INTERFACE
     DOUBLE PRECISION FUNCTION PROPFUNC(STATE, REACTION, PARAMETERS)
       IMPLICIT NONE
       INTEGER, INTENT(IN) :: STATE(:), REACTION
       DOUBLE PRECISION, INTENT(IN), OPTIONAL :: PARAMETERS(:)
     END FUNCTION PROPFUNC
  END INTERFACE

  TYPE :: CME_MODEL
     LOGICAL :: LOADED = .FALSE.
     INTEGER :: NSPECIES = 0
     INTEGER :: NREACTIONS = 0
     INTEGER :: NPARAMETERS = 0
     DOUBLE PRECISION, DIMENSION(:), ALLOCATABLE :: PARAMETER_VAL
     ! one column per reaction
     INTEGER, DIMENSION(:, :), ALLOCATABLE :: STOICHIOMETRY
     CHARACTER(LEN=20), ALLOCATABLE, DIMENSION(:) :: SPECIES_NAMES, PARAMETER_NAMES
     TYPE(EXPRESSION), DIMENSION(:), ALLOCATABLE, PRIVATE :: PROPEXPR
     PROCEDURE(PROPFUNC), POINTER, NOPASS :: CUSTOMPROP => NULL()
   CONTAINS
     PROCEDURE :: CREATE
     PROCEDURE :: LOAD
     PROCEDURE :: RESET_PARAMETERS
     PROCEDURE :: PROPENSITY => PROPENSITY_BUILTIN
     PROCEDURE :: EXPORT_PROGRAM
  END TYPE CME_MODEL

  PRIVATE :: UPPER, FIRST_TOKEN, NEXT_LINE, PARSE_REACTION

CONTAINS

  SUBROUTINE CREATE(THIS, N_SPECIES, N_REACTIONS, N_PARAMETERS)
    CLASS(CME_MODEL) :: THIS
    INTEGER :: N_SPECIES, N_REACTIONS, N_PARAMETERS
    THIS%NSPECIES = N_SPECIES
    THIS%NREACTIONS = N_REACTIONS
    THIS%NPARAMETERS = N_PARAMETERS
    IF (ALLOCATED(THIS%STOICHIOMETRY)) DEALLOCATE(THIS%STOICHIOMETRY)
    IF (ALLOCATED(THIS%PARAMETER_VAL)) DEALLOCATE(THIS%PARAMETER_VAL)
    ALLOCATE(THIS%STOICHIOMETRY(N_SPECIES, N_REACTIONS), THIS%PARAMETER_VAL(N_PARAMETERS))
    THIS%STOICHIOMETRY = 0
    THIS%PARAMETER_VAL = 0.0D0
  END SUBROUTINE CREATE

  ! .input format (SURVEY.md appendix B.4): a keyword alone on a line opens a
  ! section - nspecies/nreactions/nparameters (one integer), species /
  ! parameters (one name per line), reactions ("lhs -> rhs" or "<-", terms
  ! separated by " + ", optional integer coefficient as in 2M, "0" = nothing),
  ! propensities (one expression per reaction, in reaction order).
  SUBROUTINE LOAD(THIS, FILENAME)
    CLASS(CME_MODEL) :: THIS
    CHARACTER(LEN=*), OPTIONAL :: FILENAME
    CHARACTER(LEN=400) :: LINE
    CHARACTER(LEN=20), ALLOCATABLE :: VARS(:)
    INTEGER :: U, IOS, I
    LOGICAL :: EOF, HAVE_NS, HAVE_NR, HAVE_NP, HAVE_SPECIES, HAVE_PARAMS

    IF (PRESENT(FILENAME)) THEN
       OPEN(NEWUNIT=U, FILE=FILENAME, IOSTAT=IOS, STATUS='OLD', ACTION='READ')
    ELSE
       OPEN(NEWUNIT=U, FILE='MODEL.INPUT', IOSTAT=IOS, STATUS='OLD', ACTION='READ')
    ENDIF
    IF (IOS /= 0) STOP 'ERROR OPENING FILE '

    HAVE_NS = .FALSE.; HAVE_NR = .FALSE.; HAVE_NP = .FALSE.
    HAVE_SPECIES = .FALSE.; HAVE_PARAMS = .FALSE.
    DO
       CALL NEXT_LINE(U, LINE, EOF)
       IF (EOF) EXIT
       SELECT CASE (UPPER(FIRST_TOKEN(LINE)))
       CASE ('NSPECIES')
          CALL NEXT_LINE(U, LINE, EOF)
          READ(LINE, *, IOSTAT=IOS) THIS%NSPECIES
          HAVE_NS = IOS == 0
       CASE ('NREACTIONS')
          CALL NEXT_LINE(U, LINE, EOF)
          READ(LINE, *, IOSTAT=IOS) THIS%NREACTIONS
          HAVE_NR = IOS == 0
       CASE ('NPARAMETERS')
          CALL NEXT_LINE(U, LINE, EOF)
          READ(LINE, *, IOSTAT=IOS) THIS%NPARAMETERS
          HAVE_NP = IOS == 0
       CASE ('SPECIES')
          IF (.NOT. HAVE_NS) STOP 'MODEL INPUT ERROR: NUMBER OF SPECIES NOT DECLARED.'
          IF (ALLOCATED(THIS%SPECIES_NAMES)) DEALLOCATE(THIS%SPECIES_NAMES)
          ALLOCATE(THIS%SPECIES_NAMES(THIS%NSPECIES))
          DO I = 1, THIS%NSPECIES
             CALL NEXT_LINE(U, LINE, EOF)
             THIS%SPECIES_NAMES(I) = FIRST_TOKEN(LINE)
          ENDDO
          HAVE_SPECIES = .TRUE.
       CASE ('PARAMETERS')
          IF (.NOT. HAVE_NP) STOP 'MODEL INPUT ERROR: NUMBER OF PARAMETERS NOT DECLARED BEFORE SPECIFYING PARAMETER NAMES.'
          IF (ALLOCATED(THIS%PARAMETER_NAMES)) DEALLOCATE(THIS%PARAMETER_NAMES)
          IF (ALLOCATED(THIS%PARAMETER_VAL)) DEALLOCATE(THIS%PARAMETER_VAL)
          ALLOCATE(THIS%PARAMETER_NAMES(THIS%NPARAMETERS), THIS%PARAMETER_VAL(THIS%NPARAMETERS))
          THIS%PARAMETER_VAL = 0.0D0
          DO I = 1, THIS%NPARAMETERS
             CALL NEXT_LINE(U, LINE, EOF)
             THIS%PARAMETER_NAMES(I) = FIRST_TOKEN(LINE)
          ENDDO
          HAVE_PARAMS = .TRUE.
       CASE ('REACTIONS')
          IF (.NOT. HAVE_SPECIES) STOP 'MODEL INPUT ERROR: REACTIONS STATED BEFORE SPECIES NAMES ARE DECLARED.'
          IF (.NOT. HAVE_NS) STOP 'MODEL INPUT ERROR: NUMBER OF SPECIES NOT DECLARED.'
          IF (.NOT. HAVE_NR) STOP 'MODEL INPUT ERROR: NUMBER OF REACTIONS NOT DECLARED.'
          IF (ALLOCATED(THIS%STOICHIOMETRY)) DEALLOCATE(THIS%STOICHIOMETRY)
          ALLOCATE(THIS%STOICHIOMETRY(THIS%NSPECIES, THIS%NREACTIONS))
          DO I = 1, THIS%NREACTIONS
             CALL NEXT_LINE(U, LINE, EOF)
             CALL PARSE_REACTION(TRIM(LINE), THIS%SPECIES_NAMES, THIS%STOICHIOMETRY(:, I))
          ENDDO
       CASE ('PROPENSITIES')
          IF (.NOT. (HAVE_SPECIES .AND. HAVE_PARAMS)) &
               STOP 'MODEL INPUT ERROR: PROPENSITIES SPECIFIED BEFORE ALL SPECIES AND PARAMETERS ARE NAMED.'
          ALLOCATE(VARS(THIS%NSPECIES + THIS%NPARAMETERS))
          VARS(1:THIS%NSPECIES) = THIS%SPECIES_NAMES
          VARS(THIS%NSPECIES + 1:) = THIS%PARAMETER_NAMES
          IF (ALLOCATED(THIS%PROPEXPR)) DEALLOCATE(THIS%PROPEXPR)
          ALLOCATE(THIS%PROPEXPR(THIS%NREACTIONS))
          DO I = 1, THIS%NREACTIONS
             CALL NEXT_LINE(U, LINE, EOF)
             CALL EXPR_COMPILE(THIS%PROPEXPR(I), TRIM(LINE), VARS)
             IF (.NOT. THIS%PROPEXPR(I)%VALID) THEN
                PRINT *, 'MODEL INPUT ERROR: CANNOT PARSE PROPENSITY ', TRIM(LINE)
                STOP 1
             ENDIF
          ENDDO
          DEALLOCATE(VARS)
       CASE DEFAULT
          CONTINUE
       END SELECT
    ENDDO
    CLOSE(U)
    THIS%LOADED = .TRUE.
  END SUBROUTINE LOAD

  ! a_k(x): the compiled-in function when one is attached, else the parsed
  ! expression over (species counts, parameter values)
  DOUBLE PRECISION FUNCTION PROPENSITY_BUILTIN(THIS, STATE, REACTION)
    CLASS(CME_MODEL), INTENT(IN) :: THIS
    INTEGER, INTENT(IN) :: STATE(:)
    INTEGER, INTENT(IN) :: REACTION
    DOUBLE PRECISION :: VAL(THIS%NSPECIES + THIS%NPARAMETERS)
    IF (ASSOCIATED(THIS%CUSTOMPROP)) THEN
       PROPENSITY_BUILTIN = THIS%CUSTOMPROP(STATE(1:THIS%NSPECIES), REACTION, THIS%PARAMETER_VAL(1:THIS%NPARAMETERS))
    ELSE
       VAL(1:THIS%NSPECIES) = DBLE(STATE(1:THIS%NSPECIES))
       VAL(THIS%NSPECIES + 1:) = THIS%PARAMETER_VAL(1:THIS%NPARAMETERS)
       PROPENSITY_BUILTIN = EXPR_EVAL(THIS%PROPEXPR(REACTION), VAL)
    ENDIF
  END FUNCTION PROPENSITY_BUILTIN

  ! The parsed propensities as the flat postfix program of kfsp_set_propensity_program (include/kfsp.h): reaction k
  ! is CODE(CODE_OFF(k)+1 : CODE_OFF(k+1)) with immediates IMM(IMM_OFF(k)+1 : IMM_OFF(k+1)) (offsets 0-based, as the C
  ! side wants them); DEP_SPECIES(k) = the one species (0-based) the expression refers to - 0 if it refers to none -
  ! or -1 if it refers to several.  OK = .FALSE.: the model evaluates a compiled-in CUSTOMPROP, there is no code.
  SUBROUTINE EXPORT_PROGRAM(THIS, OK, CODE_OFF, CODE, IMM_OFF, IMM, DEP_SPECIES)
    CLASS(CME_MODEL), INTENT(IN) :: THIS
    LOGICAL, INTENT(OUT) :: OK
    INTEGER, ALLOCATABLE, INTENT(OUT) :: CODE_OFF(:), CODE(:), IMM_OFF(:), DEP_SPECIES(:)
    DOUBLE PRECISION, ALLOCATABLE, INTENT(OUT) :: IMM(:)
    INTEGER :: K, J, NC, NI, V, S
    OK = .NOT. ASSOCIATED(THIS%CUSTOMPROP) .AND. ALLOCATED(THIS%PROPEXPR)
    IF (.NOT. OK) RETURN
    ALLOCATE(CODE_OFF(THIS%NREACTIONS + 1), IMM_OFF(THIS%NREACTIONS + 1), DEP_SPECIES(THIS%NREACTIONS))
    NC = 0
    NI = 0
    DO K = 1, THIS%NREACTIONS
       CODE_OFF(K) = NC
       IMM_OFF(K) = NI
       IF (THIS%PROPEXPR(K)%VALID) THEN            ! (an expression that did not parse evaluates to 0: no code)
          NC = NC + THIS%PROPEXPR(K)%NCODE
          NI = NI + COUNT(THIS%PROPEXPR(K)%CODE(1:THIS%PROPEXPR(K)%NCODE) == 1)
       ENDIF
    ENDDO
    CODE_OFF(THIS%NREACTIONS + 1) = NC
    IMM_OFF(THIS%NREACTIONS + 1) = NI
    ALLOCATE(CODE(MAX(NC, 1)), IMM(MAX(NI, 1)))
    CODE = 0
    IMM = 0.0D0
    DO K = 1, THIS%NREACTIONS
       S = -2                                      ! -2: no species seen yet
       IF (THIS%PROPEXPR(K)%VALID) THEN
          DO J = 1, THIS%PROPEXPR(K)%NCODE
             V = THIS%PROPEXPR(K)%CODE(J)
             CODE(CODE_OFF(K) + J) = V
             IF (V > 100 .AND. V <= 100 + THIS%NSPECIES) THEN
                IF (S == -2) THEN
                   S = V - 101
                ELSEIF (S /= V - 101) THEN
                   S = -1
                ENDIF
             ENDIF
          ENDDO
          J = IMM_OFF(K + 1) - IMM_OFF(K)
          IF (J > 0) IMM(IMM_OFF(K) + 1:IMM_OFF(K) + J) = THIS%PROPEXPR(K)%IMM(1:J)
       ENDIF
       IF (S == -2) S = 0
       DEP_SPECIES(K) = S
    ENDDO
  END SUBROUTINE EXPORT_PROGRAM

  SUBROUTINE RESET_PARAMETERS(THIS, PVAL)
    CLASS(CME_MODEL) :: THIS
    DOUBLE PRECISION, DIMENSION(:) :: PVAL
    THIS%PARAMETER_VAL(1:THIS%NPARAMETERS) = PVAL(1:THIS%NPARAMETERS)
  END SUBROUTINE RESET_PARAMETERS

  ! ---------------------------------------------------------------- helpers

  ! next non-blank line of the unit
  SUBROUTINE NEXT_LINE(U, LINE, EOF)
    INTEGER, INTENT(IN) :: U
    CHARACTER(LEN=*), INTENT(OUT) :: LINE
    LOGICAL, INTENT(OUT) :: EOF
    INTEGER :: IOS, I
    EOF = .FALSE.
    DO
       READ(U, '(A)', IOSTAT=IOS) LINE
       IF (IOS /= 0) THEN
          EOF = .TRUE.
          LINE = ''
          RETURN
       ENDIF
       DO I = 1, LEN(LINE)                 ! tabs and carriage returns count as blanks
          IF (LINE(I:I) == ACHAR(9) .OR. LINE(I:I) == ACHAR(13)) LINE(I:I) = ' '
       ENDDO
       IF (LEN_TRIM(LINE) > 0) RETURN
    ENDDO
  END SUBROUTINE NEXT_LINE

  FUNCTION FIRST_TOKEN(LINE) RESULT(T)
    CHARACTER(LEN=*), INTENT(IN) :: LINE
    CHARACTER(LEN=:), ALLOCATABLE :: T
    CHARACTER(LEN=LEN(LINE)) :: L
    INTEGER :: K
    L = ADJUSTL(LINE)
    K = SCAN(L, ' ,')
    IF (K == 0) THEN
       T = TRIM(L)
    ELSE
       T = L(1:K - 1)
    ENDIF
  END FUNCTION FIRST_TOKEN

  FUNCTION UPPER(T) RESULT(R)
    CHARACTER(LEN=*), INTENT(IN) :: T
    CHARACTER(LEN=LEN(T)) :: R
    INTEGER :: I
    R = T
    DO I = 1, LEN(T)
       IF (T(I:I) >= 'a' .AND. T(I:I) <= 'z') R(I:I) = ACHAR(IACHAR(T(I:I)) - 32)
    ENDDO
  END FUNCTION UPPER

  ! "2M + D -> DNA.D" -> net change of every species
  SUBROUTINE PARSE_REACTION(STR, NAMES, NU)
    CHARACTER(LEN=*), INTENT(IN) :: STR
    CHARACTER(LEN=*), INTENT(IN) :: NAMES(:)
    INTEGER, INTENT(OUT) :: NU(:)
    INTEGER :: POS, N, E, SIDE, DIR, COEF, K, J, IOS
    CHARACTER(LEN=:), ALLOCATABLE :: TOK
    NU = 0
    SIDE = -1            ! -1: consumed (left of the arrow), +1: produced
    DIR = 0
    N = LEN_TRIM(STR)
    POS = 1
    DO WHILE (POS <= N)
       IF (STR(POS:POS) == ' ') THEN
          POS = POS + 1
          CYCLE
       ENDIF
       E = POS
       DO WHILE (E < N)
          IF (STR(E + 1:E + 1) == ' ') EXIT
          E = E + 1
       ENDDO
       TOK = STR(POS:E)
       POS = E + 1
       IF (TOK == '->') THEN
          DIR = 1
          SIDE = 1
       ELSEIF (TOK == '<-') THEN
          DIR = 2
          SIDE = 1
       ELSEIF (TOK == '+' .OR. TOK == '0') THEN
          CYCLE
       ELSE
          K = VERIFY(TOK, '0123456789')          ! first non-digit
          COEF = 1
          IF (K > 1) THEN
             READ(TOK(1:K - 1), *, IOSTAT=IOS) COEF
             IF (IOS /= 0) COEF = 1
          ENDIF
          IF (K == 0) CYCLE
          DO J = 1, SIZE(NAMES)
             IF (TOK(K:) == TRIM(NAMES(J))) THEN
                NU(J) = NU(J) + SIDE * COEF
                EXIT
             ENDIF
          ENDDO
          IF (J > SIZE(NAMES)) PRINT *, 'WARNING: SPECIES ', TOK, ' NOT DEFINED IN THE MODEL.'
       ENDIF
    ENDDO
    IF (DIR == 0) STOP 'SYNTAX ERROR IN CHEMICAL REACTION, ONLY ONE SIDE WAS WRITTEN.'
    IF (DIR == 2) NU = -NU
  END SUBROUTINE PARSE_REACTION

END MODULE MODELMODULE
