"""State space constants (mirror of reference splitp/constants.py:7-9).

The digit values drive every row/column index of a flattening
(splitp/constructions.py:166-171), so the order A, C, G, T -> 0, 1, 2, 3 is part of the
drop-in contract."""

DNA_state_space = ("A", "C", "G", "T")
DNA_state_space_dict = {state: index for index, state in enumerate(DNA_state_space)}
binary_state_space = (0, 1)
