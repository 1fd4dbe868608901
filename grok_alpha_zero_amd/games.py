"""Host-side Game plugins with the reference's Game API (Guide.py:79-283): attributes `board` (int8), `next_player`
(first = -1), `action_history`, `policy_shape`; methods get_next_player / get_legal_actions / do_action /
get_input_state / check_win / compute_policy_improvement / augment_sample and the static *_MCTS twins.

The engine runs the rules on the GPU (csrc/games.hpp); these numpy classes exist for the host side of the path
(rebuilding input states and augmentations of finished games for the replay buffer, conformance tests) and so
that a user's own reference classes (Connect4/Connect4.py:184-446, Gomoku/Gomoku.py:87-303,
TicTacToe/Tictactoe.py:151-359) can be swapped in unchanged — `self_play.run_self_play` only calls the API above.
Behavioural quirks of the reference that reach the replay buffer are kept (cited inline).
"""
import numpy as np


def _line_win(board, y, x, player, k):
    """Does the stone at (y, x) belong to a straight run of >= k stones of `player`?"""
    H, W = board.shape
    for dy, dx in ((0, 1), (1, 0), (1, 1), (1, -1)):
        run = 1
        for sgn in (1, -1):
            yy, xx = y + sgn * dy, x + sgn * dx
            while 0 <= yy < H and 0 <= xx < W and board[yy, xx] == player:
                run += 1
                yy += sgn * dy
                xx += sgn * dx
        if run >= k:
            return True
    return False


class _GridGame:
    H = W = K = 0
    ENGINE_NAME = ""

    def __init__(self):
        self.board = np.zeros((self.H, self.W), np.int8)
        self.next_player = -1
        self.action_history = []
        self.policy_shape = (self.H * self.W,)

    def get_next_player(self):
        return self.next_player

    def get_legal_actions(self):
        return self.get_legal_actions_MCTS(self.board, -self.next_player, np.array(self.action_history))

    def get_input_state(self):
        return self.get_input_state_MCTS(self.board, -self.next_player, np.array(self.action_history))

    def check_win(self):
        return self.check_win_MCTS(self.board, -self.next_player, np.array(self.action_history))

    # engine <-> plugin action coding: the engine carries one int per action
    @classmethod
    def action_to_index(cls, action):
        x, y = int(action[0]), int(action[1])
        return y * cls.W + x

    @classmethod
    def index_to_action(cls, idx):
        return np.array([idx % cls.W, idx // cls.W])

    # ---- cell games (TicTacToe / Gomoku): action = [x, y]
    @staticmethod
    def _legal_cells(board):
        return np.argwhere(board == 0)[:, ::-1]

    def do_action(self, action):
        x, y = int(action[0]), int(action[1])
        if self.board[y, x] != 0:
            raise ValueError("Illegal move")
        self.board[y, x] = self.next_player
        self.next_player = -self.next_player
        self.action_history.append(np.array(action))

    def compute_policy_improvement(self, statistics):
        pol = np.zeros((self.H, self.W), np.float32)
        for (x, y), prob in statistics:
            pol[int(y), int(x)] = prob
        return pol.reshape(-1)

    def augment_sample(self, input_states, policies):
        """The 8 board symmetries, in the reference's order: id, flipud, fliplr, rot90, flipud(rot90),
        fliplr(rot90), rot180, rot270 (Tictactoe.py:321-358, Gomoku.py:265-303) -> [8, T, ...]."""
        T = input_states.shape[0]
        pol = np.asarray(policies, np.float32).reshape(T, self.H, self.W)
        st = np.asarray(input_states)

        def sym(a):
            r1 = np.rot90(a, 1, axes=(1, 2))
            return [a, a[:, ::-1], a[:, :, ::-1], r1, r1[:, ::-1], r1[:, :, ::-1], np.rot90(a, 2, axes=(1, 2)), np.rot90(a, 3, axes=(1, 2))]
        return (np.stack(sym(st)).astype(self.board.dtype), np.stack(sym(pol)).reshape(8, T, -1).astype(np.float32))


class TicTacToe(_GridGame):
    H = W = 3
    K = 3
    ENGINE_NAME = "TicTacToe"

    @staticmethod
    def get_legal_actions_MCTS(board, current_player, action_history):
        return _GridGame._legal_cells(board)

    @staticmethod
    def get_legal_actions_policy_MCTS(board, current_player, action_history, policy, normalize=True, shuffle=False):
        legal = _GridGame._legal_cells(board)
        p = np.asarray(policy)[board.reshape(-1) == 0]
        if normalize:
            p = p / np.sum(p)
        return legal, p

    @staticmethod
    def do_action_MCTS(board, action, next_player):
        board[int(action[1]), int(action[0])] = next_player
        return board

    @staticmethod
    def get_input_state_MCTS(board, current_player, action_history):
        return np.stack((np.full_like(board, -current_player), board), -1)

    @staticmethod
    def check_win_MCTS(board, current_player, action_history):
        lines = list(board) + list(board.T) + [np.diag(board), np.diag(np.fliplr(board))]
        if any(l[0] != 0 and l[0] == l[1] == l[2] for l in lines):
            return current_player
        return 0 if np.all(board != 0) else -2


class Gomoku(_GridGame):
    H = W = 15
    K = 5
    ENGINE_NAME = "Gomoku"

    def do_action(self, action):
        super().do_action(np.array(action, np.uint8))

    @staticmethod
    def get_legal_actions_MCTS(board, current_player, action_history):
        return _GridGame._legal_cells(board).astype(np.uint8)

    @staticmethod
    def get_legal_actions_policy_MCTS(board, current_player, action_history, policy, normalize=True, shuffle=False):
        p = np.asarray(policy)[board.reshape(-1) == 0]
        if normalize:
            p = p / np.sum(p)
        return _GridGame._legal_cells(board).astype(np.uint8), p

    @staticmethod
    def do_action_MCTS(board, action, next_player):
        board[int(action[1]), int(action[0])] = next_player
        return board

    @staticmethod
    def get_input_state_MCTS(board, current_player, action_history):
        return np.stack((np.full_like(board, -current_player), board), -1)

    @staticmethod
    def check_win_MCTS(board, current_player, action_history):
        x, y = int(action_history[-1][0]), int(action_history[-1][1])
        return current_player if _line_win(board, y, x, current_player, 5) else -2     # never a draw (Gomoku.py:249-255)


class Connect4(_GridGame):
    H, W, K = 6, 7, 4
    ENGINE_NAME = "Connect4"

    def __init__(self):
        super().__init__()
        self.policy_shape = (7,)

    @classmethod
    def action_to_index(cls, action):
        return int(action)

    @classmethod
    def index_to_action(cls, idx):
        return np.int8(idx)

    def do_action(self, action):
        self.do_action_MCTS(self.board, int(action), self.next_player)
        self.next_player = -self.next_player
        self.action_history.append(action)

    def get_input_state(self):
        return self.get_input_state_MCTS(self.board, -self.next_player, np.array(self.action_history, dtype=np.int8))

    @staticmethod
    def get_legal_actions_MCTS(board, next_player, action_history):
        return np.flatnonzero(board[0] == 0).astype(np.int8)          # column not full <=> its top cell is empty

    @staticmethod
    def get_legal_actions_policy_MCTS(board, current_player, action_history, policy, normalize=True, shuffle=False):
        legal = np.flatnonzero(board[0] == 0).astype(np.int8)
        p = np.asarray(policy)[legal]
        if normalize:
            p = p / np.sum(p)
        return legal, p

    @staticmethod
    def do_action_MCTS(board, action, next_player):
        row = 5 - int(np.count_nonzero(board[:, action]))
        board[row, action] = next_player
        return board

    @staticmethod
    def get_input_state_MCTS(board, current_player, action_history):
        """Planes (current player | board-2 | board-1 | board); with >= 4 moves played plane 0 holds the board three
        moves back instead of the player plane — the reference's board_state[i - 1] at i = -3 (Connect4.py:340-345)."""
        planes = np.zeros((4, 6, 7), np.int8)
        planes[0] = current_player
        planes[3] = board
        prev = board.copy()
        for i in range(1, min(len(action_history) - 1, 3) + 1):
            x = int(action_history[-i])
            prev[np.flatnonzero(prev[:, x])[0], x] = 0
            planes[3 - i] = prev
        return np.transpose(planes, (1, 2, 0))

    @staticmethod
    def check_win_MCTS(board, current_player, action_history):
        x = int(action_history[-1])
        y = int(np.flatnonzero(board[:, x] == current_player)[0])
        if _line_win(board, y, x, current_player, 4):
            return current_player
        return 0 if np.all(board != 0) else -2

    def compute_policy_improvement(self, statistics):
        pol = np.zeros(7, np.float32)
        for action, prob in statistics:
            pol[int(action)] = prob
        return pol

    @staticmethod
    def augment_sample(board, policy):
        """[identity, np.fliplr] on BOTH arrays, as the reference does: on the [T,6,7,4] states np.fliplr reverses
        axis 1 = board ROWS, while the [T,7] policy is mirrored along columns (Connect4.py:442-443) — kept as is."""
        return np.stack((board, np.fliplr(board))), np.stack((policy, np.fliplr(policy)))


GAMES = {"TicTacToe": TicTacToe, "Connect4": Connect4, "Gomoku": Gomoku}
